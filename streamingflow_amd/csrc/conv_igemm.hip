// Implicit-GEMM convolution on the CDNA4 fp32 matrix cores (v_mfma_f32_16x16x4_f32) with fused
// conv-GRU / LayerNorm / trusting-gate / Gaussian-sample epilogues.  gfx950 only.
//
//   D[cout][pixel] = sum_k  W[cout][k] * X_im2col[k][pixel],   k = (tap, channel)
//
// * A operand = packed weights, B operand = activations gathered tap by tap (im2col on the fly,
//   zero padding, optional channel concat of two tensors, optional reset-gate multiply
//   (1-r)*s, optional per-channel SE scale, optional nearest x2 upsampling on read).
// * One wave owns MT x NT tiles of 16(cout) x 16(pixel); a workgroup is WM x WN such waves times
//   KS "K-groups" that each take every KS-th 32-deep K chunk (in-workgroup split-K, reduced
//   through LDS in a fixed order => bitwise reproducible).  Small BEV latents (50x50) have far
//   fewer output tiles than the chip has SIMDs; split-K inside the workgroup is what puts
//   4-8 waves on every busy CU without a second reduction launch.
// * Operands are staged global -> registers -> LDS (issue-early / write-late, one barrier per
//   chunk, double buffered).  LDS rows are 36 floats (32 + 4 pad): a wave's ds_read_b64
//   fragment reads (row = lane&15, k = 8t + 2*(lane>>4)) then hit 64 distinct banks.
// * Accumulator layout (16x16x4): lane l holds D[cout = 4*(l>>4)+r][pixel = l&15] in register r,
//   i.e. four consecutive channels of one pixel => NHWC float4 stores, and per-pixel channel
//   reductions (LayerNorm, 1x1->2 logits) are in-register sums + two xor-shuffles (16, 32).
#include "sf_math.h"

#include <type_traits>
#include <cstdlib>

namespace sf {

#ifdef SF_STAMP
__device__ unsigned long long* g_sf_stamps = nullptr;
hipError_t set_stamp_buffer(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sf_stamps), &p, sizeof(p)); }
#else
hipError_t set_stamp_buffer(unsigned long long*) { return hipErrorNotSupported; }
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_LRELU:   return v > 0.f ? v : 0.1f * v;
    case ACT_RELU:    return v > 0.f ? v : 0.f;
    case ACT_TANH:    return tanhf(v);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_GELU:    return 0.5f * v * (1.f + spm_erf(v * 0.70710678118654752440f));
    default:          return v;
  }
}

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + spm_erf(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// sum over the 4 lanes (l, l^16, l^32, l^48) that hold the other channels of this pixel
__device__ __forceinline__ float pix_allreduce(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// sum over the 16 pixels (lanes with equal l>>4) of a wave tile
__device__ __forceinline__ float tile_px_reduce(float v) { return spm_row16_sum(v); }

// One ds_read_b64 per fragment.  The 36-float row pitch is conflict-free for ds_read_b64 (banks
// (a/4) mod 64 over 32-lane halves) but 2-way conflicting for the ds_read2_b64 hipcc would merge
// two of these into (banks (a/4) mod 32 over 16-lane groups; measured: SQ_LDS_BANK_CONFLICT = 40 %
// of SQ_LDS_IDX_ACTIVE) — volatile keeps the loads separate.
__device__ __forceinline__ float2 lds_read_b64(const float* p) {
  typedef const volatile __attribute__((address_space(3))) unsigned long long lds_u64;
  const unsigned long long u = *(lds_u64*)p;   // explicit LDS address space: ds_read_b64, not flat_load
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}

// ---- split-bf16 operands (opt-in math mode "bf16x3", never the default) --------------------------------------------------
// a = hi + lo with hi = bf16(a), lo = bf16(a - hi) (round to nearest even): a * b ~ hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulators — products of bf16 values are exact in fp32, the dropped lo*lo term and
// the residual of the split are ~2^-17 relative (profiles/r03_bf16x3_accuracy_study.json: <= 5e-5 max-abs on the BEV logits
// of the full-size forward, against 2e-2 for plain bf16).  Weights are split once by sf_pack_conv (SF_PACK_BF16X3: every
// aligned group of 8 K values is stored as [8 x bf16 hi][8 x bf16 lo], the same 32 bytes, so the staging DMAs do not
// change); activations stay fp32 in memory and in LDS and are split in registers after the fragment read.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
__device__ __forceinline__ f32x4 lds_read_b128(const float* p) {
  typedef const __attribute__((address_space(3))) f32x4 lds_f4;
  return *(lds_f4*)p;
}
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: a in the low half
  const bf16x2 t = __builtin_convertvector((f32x2){a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, t);
}
// eight consecutive K values (two float4) -> their bf16 hi and lo pieces, K ascending from the low half of word 0
__device__ __forceinline__ void split_bf16x8(const f32x4 x0, const f32x4 x1, bf16x8& hi, bf16x8& lo) {
  u32x4v h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = i < 2 ? x0[2 * i] : x1[2 * i - 4], b = i < 2 ? x0[2 * i + 1] : x1[2 * i - 3];
    const unsigned pk = pk_bf16(a, b);
    h[i] = pk;
    l[i] = pk_bf16(a - __uint_as_float(pk << 16), b - __uint_as_float(pk & 0xffff0000u));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

#ifndef SF_SETPRIO
#define SF_SETPRIO 1
#endif
constexpr bool SETPRIO = SF_SETPRIO;
// log2 of the XCD tile chunk of the LDS-DMA kernel's large launches (SF_XCD_CHUNK = 32 tiles; 0 switches the order off)
static int xcd_chunk_log2() {
  static const int v = [] {
    const char* e = std::getenv("SF_XCD_CHUNK");
    int c = e ? std::atoi(e) : 32, l = -1;
    while (c > 0) { ++l; c >>= 1; }
    return l;
  }();
  return v;
}
#ifndef SF_GDIAG
#define SF_GDIAG 0  // same for the LDS-DMA kernel: 1 = weights only, 2 = pixels only, 3 = every chunk re-reads chunk 0, 4 = no DMA
#endif
#ifndef SF_DIAG
#define SF_DIAG 0   // timing probes of the staged main loop (tools/experiments/diag_loop.sh); 0 = the product
#endif
constexpr int LDS_ROW = 36;   // floats per staged row: 32 K values + 4 pad
constexpr int BK = 32;

// ---- epilogues (shared by the LDS-staged and the direct-fragment kernels) -------------------
// acc[m][n]: 16x16 tiles, lane (j = lane&15, g = lane>>4) holds channels m0 + 16m + 4g + (0..3) of
// pixel p0 + 16n + j.  LayerNorm epilogues need m0 == 0 and 16*MT >= cout (all channels in-wave).
template <int MT, int NT, int EPI>
__device__ __forceinline__ void run_epilogue(const ConvProblem& P, f32x4 (&acc)[MT][NT], const int m0, const int p0,
                                             const int lane, const int Ptot, const int HWout) {
  const int j = lane & 15, g = lane >> 4;

  if constexpr (EPI == EPI_AFFINE || EPI == EPI_BLEND) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int gp = p0 + n * 16 + j;
      const bool pv = gp < Ptot;
      const int img = pv ? gp / HWout : 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int c = m0 + m * 16 + 4 * g;
        const bool cv = c < P.cout;
        float4 y = zero4();
        if (pv && cv) {
          float4 sc = P.scale ? ld4(P.scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
          float4 bi = P.bias ? ld4(P.bias + (P.bias_per_img ? (size_t)img * P.cout : 0) + c) : zero4();
          float4 v;
          v.x = acc[m][n][0] * sc.x + bi.x; v.y = acc[m][n][1] * sc.y + bi.y;
          v.z = acc[m][n][2] * sc.z + bi.z; v.w = acc[m][n][3] * sc.w + bi.w;
          if constexpr (EPI == EPI_AFFINE) {
            const bool act_last = (P.mode & 2) != 0;   // ResNet BasicBlock: relu(bn(conv) + identity)
            y = v;
            if (!act_last) {
              y.x = act_apply(v.x, P.act); y.y = act_apply(v.y, P.act);
              y.z = act_apply(v.z, P.act); y.w = act_apply(v.w, P.act);
            }
            if (P.clamp_from >= 0) {   // clamp(log_sigma) of the distribution heads (motion_modules.py:44,85)
              if (c + 0 >= P.clamp_from) y.x = fminf(fmaxf(y.x, P.clamp_lo), P.clamp_hi);
              if (c + 1 >= P.clamp_from) y.y = fminf(fmaxf(y.y, P.clamp_lo), P.clamp_hi);
              if (c + 2 >= P.clamp_from) y.z = fminf(fmaxf(y.z, P.clamp_lo), P.clamp_hi);
              if (c + 3 >= P.clamp_from) y.w = fminf(fmaxf(y.w, P.clamp_lo), P.clamp_hi);
            }
            if (P.add) {
              float4 ad = ld4(P.add + (size_t)gp * P.add_cs + c);
              if (P.add_scale) {
                float4 as = ld4(P.add_scale + (size_t)img * P.cout + c);
                ad.x *= as.x; ad.y *= as.y; ad.z *= as.z; ad.w *= as.w;
              }
              y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w;
            }
            if (act_last) {
              y.x = act_apply(y.x, P.act); y.y = act_apply(y.y, P.act);
              y.z = act_apply(y.z, P.act); y.w = act_apply(y.w, P.act);
            }
            if (P.out2 && c >= P.gate_from) {   // GRU gates, reset half: also emit (1 - r) * s, the candidate conv's input
              const int cg = c - P.gate_from;
              const float4 sv = ld4(P.e1 + (size_t)gp * P.e1_cs + cg);
              st4(P.out2 + (size_t)gp * P.out2_cs + cg,
                  make_float4(sv.x * (1.f - y.x), sv.y * (1.f - y.y), sv.z * (1.f - y.z), sv.w * (1.f - y.w)));
            }
          } else {  // EPI_BLEND  (temporal.py:56, temporal_ode_bayes.py:145,160; BEVerse cells apply BN+ReLU first)
            v.x = act_apply(v.x, P.act); v.y = act_apply(v.y, P.act);
            v.z = act_apply(v.z, P.act); v.w = act_apply(v.w, P.act);
            float4 u = ld4(P.e0 + (size_t)gp * P.e0_cs + c);
            float4 s = ld4(P.e1 + (size_t)gp * P.e1_cs + c);
            if (P.mode & 1) {   // SpatialGRUODECell (temporal_ode_bayes.py:60): dh = u * (h~ - s)
              y.x = u.x * (v.x - s.x); y.y = u.y * (v.y - s.y); y.z = u.z * (v.z - s.z); y.w = u.w * (v.w - s.w);
            } else {
              y.x = (1.f - u.x) * s.x + u.x * v.x; y.y = (1.f - u.y) * s.y + u.y * v.y;
              y.z = (1.f - u.z) * s.z + u.z * v.z; y.w = (1.f - u.w) * s.w + u.w * v.w;
            }
          }
          bool planar = false;
          if constexpr (EPI == EPI_AFFINE) planar = P.out_planar != 0;      // block-uniform
          if (planar) {      // 16 consecutive pixels of a channel per lane row: 64-byte runs
            float* o = P.out + (size_t)(img / P.pl_div) * P.pl_sa + (size_t)(img % P.pl_div) * P.pl_sb + (size_t)c * HWout + (gp - img * HWout);
            o[0] = y.x; o[(size_t)HWout] = y.y; o[2 * (size_t)HWout] = y.z; o[3 * (size_t)HWout] = y.w;
          } else {
            st4(P.out + (size_t)gp * P.out_cs + P.out_co + c, y);
          }
        }
        if constexpr (EPI == EPI_AFFINE) {
          if (P.chansum) {   // block-uniform branch; all lanes take part in the shuffles
            float4 s4;
            s4.x = tile_px_reduce(y.x); s4.y = tile_px_reduce(y.y);
            s4.z = tile_px_reduce(y.z); s4.w = tile_px_reduce(y.w);
            if (j == 0 && cv && (p0 + n * 16) < Ptot) {   // wave tiles past the last pixel own no slot
              int tile16 = p0 / 16 + n;
              st4(P.chansum + (size_t)tile16 * P.cout + c, s4);
            }
          }
        }
      }
    }
  }

  if constexpr (EPI == EPI_LNG || EPI == EPI_TRUST) {
    // the wave holds every channel of its pixels (m0 == 0, 16*MT >= cout; checked on the host)
    const float inv_c = 1.f / (float)P.cout;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int gp = p0 + n * 16 + j;
      const bool pv = gp < Ptot;
      float v[MT][4];
      bool cvm[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        cvm[m] = (m0 + m * 16 + 4 * g) < P.cout;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[m][q] = acc[m][n][q];
      }
      const bool do_ln = (EPI == EPI_TRUST) || (P.mode & 1);
      if (do_ln) {   // convolutions.py:303-308 (channels_first LayerNorm)
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (cvm[m]) s += (v[m][0] + v[m][1]) + (v[m][2] + v[m][3]);
        const float mean = pix_allreduce(s) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (cvm[m]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { float d = v[m][q] - mean; sq += d * d; }
          }
        const float var = pix_allreduce(sq) * inv_c;
        const float rstd = 1.f / sqrtf(var + P.eps);
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (cvm[m]) {
            const int c = m0 + m * 16 + 4 * g;
            float4 w = ld4(P.scale + c), b = ld4(P.bias + c);
            v[m][0] = w.x * ((v[m][0] - mean) * rstd) + b.x;
            v[m][1] = w.y * ((v[m][1] - mean) * rstd) + b.y;
            v[m][2] = w.z * ((v[m][2] - mean) * rstd) + b.z;
            v[m][3] = w.w * ((v[m][3] - mean) * rstd) + b.w;
          }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[m][q] = gelu_f(v[m][q]);

      if constexpr (EPI == EPI_LNG) {
        if (pv) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (cvm[m]) {
              const int c = m0 + m * 16 + 4 * g;
              float4 y = make_float4(v[m][0], v[m][1], v[m][2], v[m][3]);
              if (P.add) {   // Bottleblock residual (convolutions.py:375-380): layers(x) + x | projection(x)
                const float4 ad = ld4(P.add + (size_t)gp * P.add_cs + c);
                y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w;
              }
              st4(P.out + (size_t)gp * P.out_cs + P.out_co + c, y);
            }
        }
      } else {
        // trusting gate tail (temporal_ode_bayes.py:124-131 / :268-275, convolutions.py:375-380):
        // bb = t3 + skip ; z = W2 bb ; g = softmax(z) ; cur = r2*g0 + r1*g1
        const size_t po = (size_t)(pv ? gp : 0) * P.cout;
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (cvm[m]) {
            const int c = m0 + m * 16 + 4 * g;
            float4 sk = ld4(P.e0 + po + c);
            float4 w0 = ld4(P.e1 + c), w1 = ld4(P.e1 + P.cout + c);
            float b0 = v[m][0] + sk.x, b1 = v[m][1] + sk.y, b2 = v[m][2] + sk.z, b3 = v[m][3] + sk.w;
            z0 += (w0.x * b0 + w0.y * b1) + (w0.z * b2 + w0.w * b3);
            z1 += (w1.x * b0 + w1.y * b1) + (w1.z * b2 + w1.w * b3);
          }
        z0 = pix_allreduce(z0);
        z1 = pix_allreduce(z1);
        const float zm = fmaxf(z0, z1);
        const float ez0 = expf(z0 - zm), ez1 = expf(z1 - zm);
        const float g0 = ez0 / (ez0 + ez1), g1 = ez1 / (ez0 + ez1);
        if (pv) {
          const float* cf = P.coef ? P.coef + (size_t)(gp / HWout) * P.coef_stride : nullptr;
          const float c0f = cf ? cf[0] : 0.f;
          const float c1f = (cf && P.out2) ? cf[1] : 0.f;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (cvm[m]) {
              const int c = m0 + m * 16 + 4 * g;
              float4 r2 = ld4(P.e2 + po + c), r1 = ld4(P.e3 + po + c);
              float4 cur;
              cur.x = r2.x * g0 + r1.x * g1; cur.y = r2.y * g0 + r1.y * g1;
              cur.z = r2.z * g0 + r1.z * g1; cur.w = r2.w * g0 + r1.w * g1;
              if (P.mode & 1) {   // derivative: d = cur - s ; out = base + coef0*d
                float4 s = ld4(P.e4 + po + c), base = ld4(P.e5 + po + c);
                float4 d = make_float4(cur.x - s.x, cur.y - s.y, cur.z - s.z, cur.w - s.w);
                float4 o;
                o.x = base.x + c0f * d.x; o.y = base.y + c0f * d.y;
                o.z = base.z + c0f * d.z; o.w = base.w + c0f * d.w;
                if (P.out2) {
                  float4 b2 = (P.mode & 2) ? ld4(P.out2 + po + c) : base;
                  b2.x += c1f * d.x; b2.y += c1f * d.y; b2.z += c1f * d.z; b2.w += c1f * d.w;
                  st4(P.out2 + po + c, b2);
                }
                st4(P.out + po + c, o);
              } else {
                st4(P.out + po + c, cur);
              }
            }
        }
      }
    }
  }

  if constexpr (EPI == EPI_SAMPLE) {
    // Packed cout rows are interleaved so that a lane holds (loc c, loc c+1, raw c, raw c+1):
    // row 16*T + 4*g + r  <->  r<2: loc channel 8T+2g+r ; r>=2: raw channel 8T+2g+(r-2).
    const int Chalf = P.cout >> 1;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int gp = p0 + n * 16 + j;
      const bool pv = gp < Ptot;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = m0 + m * 16 + 4 * g;
        const int c = (row >> 4) * 8 + 2 * g;   // logical loc channel
        if (pv && c < Chalf) {
          float4 bi = P.bias ? ld4(P.bias + row) : zero4();
          float q0 = act_apply(acc[m][n][0] + bi.x, P.act), q1 = act_apply(acc[m][n][1] + bi.y, P.act);
          float q2 = act_apply(acc[m][n][2] + bi.z, P.act), q3 = act_apply(acc[m][n][3] + bi.w, P.act);
          const float2 e = P.e0 ? *reinterpret_cast<const float2*>(P.e0 + (size_t)gp * Chalf + c)
                                : spm_philox_normal2(P.philox, P.draw, (unsigned)gp, (unsigned)c);
          float2 o;
          o.x = q0 + e.x * (softplus_f(q2) + 1e-8f);     // model_utils.py:84,107-108
          o.y = q1 + e.y * (softplus_f(q3) + 1e-8f);
          *reinterpret_cast<float2*>(P.out + (size_t)gp * Chalf + c) = o;
          if (P.out2) {   // raw q parameters, reference channel order [loc | raw]
            *reinterpret_cast<float2*>(P.out2 + (size_t)gp * P.cout + c) = make_float2(q0, q1);
            *reinterpret_cast<float2*>(P.out2 + (size_t)gp * P.cout + Chalf + c) = make_float2(q2, q3);
          }
        }
      }
    }
  }
}

// ---- cross-workgroup split-K: slab publish, ticket, last arriver reduces in slice order ------
// Round 2: the partial tiles leave with sc1 (write-through) 16-byte stores, every wave drains, barrier, ONE agent-scope
// ticket per workgroup; the workgroup whose ticket is the last re-reads ALL slices with sc1 loads in slice order
// (MI355X_MICROARCH.md, hand-off table row 1).  No release / acquire fence: the round-1 form (plain stores + agent
// release + acquire) cost 5-14 us per launch in the fences (tools/r02/stamps.py); placement independent either way;
// fixed summation order => bitwise reproducible.
// Returns false for the workgroups that are done (not the last arriver of their tile).
template <int MT, int NT, int NW>
__device__ __forceinline__ bool splitk_handoff(const ConvProblem& P, f32x4 (&acc)[MT][NT], const int nsplit, const int tile,
                                               const int wave, const int lane, const int tid, float* smem) {
  constexpr int PER_WAVE = MT * NT * 4 * 64;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
  float* const tile_slab = P.slab + (size_t)tile * nsplit * NW * PER_WAVE;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_slab, (short)0, nsplit * NW * PER_WAVE * 4, 0x00020000);
  const int lane_off = (wave * PER_WAVE + lane * 4) * 4;      // bytes: [slice][wave][m*NT+n][lane][4]
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(acc[m][n][0]), __float_as_uint(acc[m][n][1]), __float_as_uint(acc[m][n][2]),
                                                     __float_as_uint(acc[m][n][3])},
                                             rs, lane_off + (m * NT + n) * 1024, (int)blockIdx.z * (NW * PER_WAVE * 4), 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = reinterpret_cast<int*>(smem);
  if (tid == 0) {
    unsigned* cnt = P.counters + tile;
    if (P.fenced) {      // reference form: what the write-through stores make unnecessary on gfx950 (MI355X_MICROARCH.md hand-off table; not an architectural guarantee)
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
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < nsplit; ++z) {      // slice order, own slice included: the sum does not depend on who arrives last
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + (m * NT + n) * 1024, z * (NW * PER_WAVE * 4), 16);
        acc[m][n][0] += __uint_as_float(t[0]); acc[m][n][1] += __uint_as_float(t[1]);
        acc[m][n][2] += __uint_as_float(t[2]); acc[m][n][3] += __uint_as_float(t[3]);
      }
  }
  return true;
#else
  (void)P; (void)acc; (void)nsplit; (void)tile; (void)wave; (void)lane; (void)tid; (void)smem;
  return true;
#endif
}

template <int MT, int NT, int WM, int WN, int KS, int EPI, int KB = 32, bool SWZ = false>
__global__ __launch_bounds__(64 * WM * WN * KS) void conv_igemm_kernel(const ConvLaunch L) {
  constexpr int BKc = KB;          // K depth of one staged chunk (32, or 64 when cin_pad % 64 == 0)
  // LDS row pitch: KB+4 floats ((KB+4)/4 odd => conflict-free ds_read_b64 fragments), or — SWZ, KB = 32
  // only — unpadded 128-B rows whose eight 16-B slots are XOR-swizzled with (row>>1)&7: equally
  // conflict-free, 32 KB instead of 36 KB per 64x64 workgroup => 5 instead of 4 workgroups per CU
  constexpr int ROW = SWZ ? KB : KB + 4;
  static_assert(!SWZ || KB == 32, "slot swizzle is defined for 32-deep chunks");
  constexpr int F4 = KB / 4;       // float4 slots per staged row
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int TG = 64 * WM * WN;        // threads per K-group
  constexpr int ROWS_PER_PASS = TG / F4;
  constexpr int A_SLOTS = BM / ROWS_PER_PASS;
  constexpr int B_SLOTS = BN / ROWS_PER_PASS;
  static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/threads mismatch");
  constexpr int GROUP_FLOATS = 2 * (BM + BN) * ROW;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const ConvProblem& P = L.p[blockIdx.y];
  const int Ptot = P.n_img * P.Hout * P.Wout;
  const int n_mt = (P.cout_pad + BM - 1) / BM;
  const int m_tile = blockIdx.x % n_mt;
  const int p_tile = blockIdx.x / n_mt;
  if (p_tile * BN >= Ptot) return;   // block-uniform

  const int tid = threadIdx.x;
  // wave-uniform by construction (TG is a multiple of 64): tell the compiler so it can use SALU
  const int kg = __builtin_amdgcn_readfirstlane(tid / TG);
  const int t = tid - kg * TG;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane & 15, g = lane >> 4;

  float* As = smem + kg * GROUP_FLOATS;          // [2][BM][ROW]
  float* Bs = As + 2 * BM * ROW;             // [2][BN][ROW]

  // ---- problem fields used in the K loop, read from the kernarg segment once ------------------
  // the input pointers are rebased to the tile's first image, so the per-tap element offsets of the K loop
  // (32 bit) only have to span the (at most two) images a tile touches, whatever the batch size
  const int img0 = P.gather ? 0 : (p_tile * BN) / (P.Hout * P.Wout);          // block-uniform
  const size_t img0_px = (size_t)img0 * P.Hin * P.Win;
  const float* const in0 = P.in0 + img0_px * P.in0_cs;
  const float* const in1 = P.in1 ? P.in1 + img0_px * P.in1_cs : nullptr;
  const float* const gate = P.gate ? P.gate + img0_px * P.gate_cs : nullptr;
  const float* const in_scale = P.in_scale;
  const float* const wbase = P.w;
  const int c0 = P.c0, c01 = P.c0 + P.c1;
  const int in0_cs = P.in0_cs, in1_cs = P.in1_cs, gate_cs = P.gate_cs, gate_co = P.gate_co;
  const int Win = P.Win, in_up = P.in_up, dil = P.dil, KW = P.KW;
  const int Hlog = P.Hin << P.in_up, Wlog = P.Win << P.in_up;
  const bool has_aux = (gate != nullptr) | (in_scale != nullptr);   // block-uniform
  const int* const gather = P.gather;
  const int KHg = P.KH;

  // ---- per-thread staging slots ----------------------------------------------------------
  const int k4 = t % F4;
  const int row0 = t / F4;
  const int HWout = P.Hout * P.Wout;
  int b_iy0[B_SLOTS], b_ix0[B_SLOTS], b_base[B_SLOTS], b_img[B_SLOTS];
#pragma unroll
  for (int i = 0; i < B_SLOTS; ++i) {
    int gp = p_tile * BN + row0 + i * ROWS_PER_PASS;
    bool v = gp < Ptot;
    int img = v ? gp / HWout : 0;
    int rem = gp - img * HWout;
    int oy = rem / P.Wout, ox = rem - oy * P.Wout;
    b_iy0[i] = v ? oy * P.stride - P.pad : -(1 << 28);   // invalid pixel: never in range
    b_ix0[i] = ox * P.stride - P.pad;
    b_base[i] = (v ? img - img0 : 0) * P.Hin * P.Win;
    b_img[i] = img;
  }
  // weight rows: clamp instead of predicating (rows >= cout_pad are never stored by any epilogue)
  size_t a_off[A_SLOTS];
#pragma unroll
  for (int i = 0; i < A_SLOTS; ++i) {
    int grow = m_tile * BM + row0 + i * ROWS_PER_PASS;
    grow = grow < P.cout_pad ? grow : P.cout_pad - 1;
    a_off[i] = (size_t)grow * P.ktot + k4 * 4;
  }

  const int kcpt = P.cin_pad / BKc;            // chunks per tap
  const int nchunks_all = P.KH * P.KW * kcpt;
  // cross-workgroup split-K: this workgroup owns chunks [cb, cb + nchunks)
  const int nsplit = P.nsplit > 1 ? P.nsplit : 1;
  if ((int)blockIdx.z >= nsplit) return;   // block-uniform
  const int cps = (nchunks_all + nsplit - 1) / nsplit;
  const int cb = (int)blockIdx.z * cps;
  const int nchunks = (nchunks_all - cb) < cps ? (nchunks_all - cb) : cps;
  const int niter = (nchunks + KS - 1) / KS;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Staging registers.  Every load below is UNCONDITIONAL (addresses are selected, never the
  // loads): a load under a divergent branch makes hipcc wait vmcnt(0) at the join, which would
  // serialise the whole prefetch.  Masks / gate / SE scale are applied when writing to LDS.
  float4 ra[A_SLOTS], rb[B_SLOTS], rx[B_SLOTS];
  int bflag[B_SLOTS];   // bit0: value valid, bit1: multiply by rx, bit2: multiply by (1 - rx)
  // per-tap cache of the gathered pixel (offset in pixels, -1 = outside): with KS == 1 the cursor
  // walks the channel chunks of one tap before moving on, so the bounds / address arithmetic is
  // paid once per tap instead of once per chunk (it was 4.7 VALU per MFMA)
  int tap_pix[B_SLOTS];
  int tap_off0[B_SLOTS], tap_off1[B_SLOTS];
  bool tap_fresh = true;

  // chunk cursor of this K-group: (tap row, tap col, channel chunk), advanced by KS per iteration
  int cur_kc = (cb + kg) % kcpt, cur_tap = (cb + kg) / kcpt;
  int cur_ty = cur_tap / KW, cur_tx = cur_tap - cur_ty * KW;

  auto load_chunk = [&](int chunk) {
    const float* wp = wbase + (size_t)chunk * BKc;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) ra[i] = ld4(wp + a_off[i]);
    const int c = cur_kc * BKc + k4 * 4;
    const bool s0 = c < c0;
    const bool s1 = (!s0) & (c < c01);
    const int cc = c - c0;
    if (KS > 1 || tap_fresh) {   // wave-uniform
      if (gather) {              // block-uniform: sparse convolution, the neighbour table replaces the arithmetic
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
          const int gp = p_tile * BN + row0 + i * ROWS_PER_PASS;
          tap_pix[i] = gather[(size_t)(gp < Ptot ? gp : 0) * KHg + cur_ty];
          if (gp >= Ptot) tap_pix[i] = -1;
        }
      } else {
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
          const int iy = b_iy0[i] + cur_ty * dil, ix = b_ix0[i] + cur_tx * dil;
          const bool in = (iy >= 0) & (iy < Hlog) & (ix >= 0) & (ix < Wlog);
          tap_pix[i] = in ? b_base[i] + (iy >> in_up) * Win + (ix >> in_up) : -1;
        }
      }
      // element offsets of the gathered pixel in both sources, once per tap (< 2^31 elements: host check)
#pragma unroll
      for (int i = 0; i < B_SLOTS; ++i) {
        const int px = tap_pix[i] >= 0 ? tap_pix[i] : 0;
        tap_off0[i] = px * in0_cs;
        tap_off1[i] = px * in1_cs - c0;
      }
    }
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
      const bool ok = (tap_pix[i] >= 0) & (s0 | s1);
      const size_t pix = ok ? (size_t)tap_pix[i] : 0;
      const int eoff = s1 ? tap_off1[i] + c : tap_off0[i] + (s0 ? c : 0);
      const float* p = (s1 ? in1 : in0) + eoff;
      rb[i] = ld4(p);
      int fl = ok ? 1 : 0;
      if (has_aux) {   // block-uniform branch, no dependent use inside
        const bool m1 = ok & s0 & (in_scale != nullptr);
        const bool m2 = ok & s1 & (gate != nullptr);
        const float* q = m1 ? in_scale + (size_t)b_img[i] * c0 + c : (m2 ? gate + pix * gate_cs + gate_co + cc : in0);
        rx[i] = ld4(q);
        fl |= (m1 ? 2 : 0) | (m2 ? 4 : 0);
      }
      bflag[i] = fl;
    }
    // advance the cursor to this K-group's next chunk
    cur_kc += KS;
    tap_fresh = false;
    while (cur_kc >= kcpt) {
      cur_kc -= kcpt;
      tap_fresh = true;
      if (++cur_tx == KW) { cur_tx = 0; ++cur_ty; }
    }
  };

  auto store_chunk = [&](int buf) {
    float* a = As + buf * BM * ROW;
    float* b = Bs + buf * BN * ROW;
    const int k4s = SWZ ? (k4 ^ ((row0 >> 1) & 7)) : k4;   // ROWS_PER_PASS is a multiple of 16: row bits 1-3 = row0's
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) st4(a + (row0 + i * ROWS_PER_PASS) * ROW + k4s * 4, ra[i]);
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
      float4 v = rb[i];
      const int fl = bflag[i];
      if (has_aux) {
        const float4 x = rx[i];
        // selects, not arithmetic masks: the dummy-address loads may hold anything
        const bool sc = fl & 2, gt = fl & 4;
        v.x *= sc ? x.x : (gt ? 1.f - x.x : 1.f); v.y *= sc ? x.y : (gt ? 1.f - x.y : 1.f);
        v.z *= sc ? x.z : (gt ? 1.f - x.z : 1.f); v.w *= sc ? x.w : (gt ? 1.f - x.w : 1.f);
      }
      if (!(fl & 1)) v = zero4();
      st4(b + (row0 + i * ROWS_PER_PASS) * ROW + k4s * 4, v);
    }
  };

  auto compute = [&](int buf) {
    const int sx = (j >> 1) & 7;   // swizzle key of this lane's rows (tile rows are multiples of 16 apart)
    auto koff = [&](int t4) { return SWZ ? 4 * ((2 * t4 + (g >> 1)) ^ sx) + ((2 * g) & 3) : 8 * t4 + 2 * g; };
    const float* a = As + buf * BM * ROW + (wm * MT * 16 + j) * ROW;
    const float* b = Bs + buf * BN * ROW + (wn * NT * 16 + j) * ROW;
    // fragments of k-group t4+1 are read while the MFMAs of k-group t4 run (two register sets,
    // statically indexed): the LDS latency is hidden inside the wave, not only by other waves
    float2 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) fa[0][m] = lds_read_b64(a + m * 16 * ROW + koff(0));
#pragma unroll
    for (int n = 0; n < NT; ++n) fb[0][n] = lds_read_b64(b + n * 16 * ROW + koff(0));
#pragma unroll
    for (int t4 = 0; t4 < BKc / 8; ++t4) {
      const int cur = t4 & 1, nxt = cur ^ 1;
      if (t4 < BKc / 8 - 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[nxt][m] = lds_read_b64(a + m * 16 * ROW + koff(t4 + 1));
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[nxt][n] = lds_read_b64(b + n * 16 * ROW + koff(t4 + 1));
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads above this group's MFMAs
      if (SETPRIO) __builtin_amdgcn_s_setprio(1);
      // all tiles with the even k first, then the odd k: MT*NT independent accumulators between
      // two MFMAs on the same one (16x16x4 f32: 32-cycle issue, 40-cycle dependent latency)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][m].x, fb[cur][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][m].y, fb[cur][n].y, acc[m][n], 0, 0, 0);
      if (SETPRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main loop: one barrier per chunk, loads of chunk i+1 in flight under MFMAs of chunk i
  if (kg < nchunks) {
    load_chunk(cb + kg);
    store_chunk(0);
  }
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const int nxt = (it + 1) * KS + kg;
    const bool has_next = nxt < nchunks;
#if SF_DIAG == 0
    if (has_next) load_chunk(cb + nxt);
    if (it * KS + kg < nchunks) compute(it & 1);
    if (has_next) store_chunk((it + 1) & 1);
    __syncthreads();
#elif SF_DIAG == 1   // MFMA + fragment reads + barrier, no staging (results meaningless: timing probe only)
    if (it * KS + kg < nchunks) compute(it & 1);
    __syncthreads();
#elif SF_DIAG == 2   // MFMA + fragment reads, no barrier
    if (it * KS + kg < nchunks) compute(it & 1);
#elif SF_DIAG == 4   // global loads issued, never written to LDS
    if (has_next) load_chunk(cb + nxt);
    if (it * KS + kg < nchunks) compute(it & 1);
    __syncthreads();
#elif SF_DIAG == 5   // LDS writes of stale registers, no global loads
    if (it * KS + kg < nchunks) compute(it & 1);
    if (has_next) store_chunk((it + 1) & 1);
    __syncthreads();
#endif
  }
#if SF_DIAG == 4
  if (niter < 0) store_chunk(0);   // keeps the staged registers alive
#endif

  // ---- in-workgroup split-K reduction (fixed order) ----------------------------------------
  if constexpr (KS > 1) {
    float* red = smem;   // [(KS-1)][WM*WN][MT*NT*4][64]
    constexpr int PER_WAVE = MT * NT * 4 * 64;
    if (kg > 0) {
      float* r = red + ((kg - 1) * (WM * WN) + wave) * PER_WAVE + lane;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) r[((m * NT + n) * 4 + q) * 64] = acc[m][n][q];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int s = 1; s < KS; ++s) {
      const float* r = red + ((s - 1) * (WM * WN) + wave) * PER_WAVE + lane;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[m][n][q] += r[((m * NT + n) * 4 + q) * 64];
    }
  }

  if (nsplit > 1 && !splitk_handoff<MT, NT, WM * WN>(P, acc, nsplit, (int)blockIdx.x, wave, lane, tid, smem)) return;   // block-uniform

  // ---- epilogue --------------------------------------------------------------------------------
  run_epilogue<MT, NT, EPI>(P, acc, m_tile * BM + wm * MT * 16, p_tile * BN + wn * NT * 16, lane, Ptot, HWout);
}

// ---- LDS-DMA kernel: the shipped implicit-GEMM kernel for every layer whose operands can be staged by a DMA ----------
// Same tiles and MFMA order as conv_igemm_kernel, but the operands go global -> LDS with buffer_load_dwordx4 ... lds:
// no VGPR round trip, no ds_write, no staging registers.  NB (2) unpadded LDS buffers of a 32-deep chunk.  An LDS-DMA
// instruction writes 64 x 16 B = eight 128-B rows contiguously, so rows are unpadded and the eight 16-B slots of a row
// are XOR-swizzled with (row >> 1) & 7 — applied to the per-lane SOURCE address.  Zero padding / out-of-range lanes
// fail the buffer range check (a DMA cannot be masked).  Handled here: channel concat of two inputs (multiples of 32
// channels) or one input of any width, stride, dilation, nearest x2 upsampling on read, the neighbour table of a sparse
// convolution, cross-workgroup split-K, and (SCALE instantiation) a per-(image, channel) SE scale applied to the pixel
// fragments after the LDS read.  A reset-gate multiply is not: the GRU gates launch writes (1 - r) * s instead.
// (tools/experiments/diag_loop.sh: without its staging the register-staged loop runs at 134 instead of 115 TFLOP/s on a
// 7-frame 128->128 layer — global loads cost 10 %, the LDS writes 6 %; this kernel reaches 126, 134 at 224 frames.)
// bf16x3 staging depth.  SF_B3_DEEP=0 (shipped): two buffers and as many workgroups per CU as fit (2 for the 128 x 128 tiles, 3 for
// 64 x 128).  -DSF_B3_DEEP=1 (experiment): as many 32-deep buffers as ~150 KB of LDS hold, one workgroup per CU, NB - 1 chunks in
// flight — measured 186.5 ms per batch-32 forward against 137.3: the loop is not short of bytes in flight, it is short of other
// workgroups to run while one waits at its per-chunk barrier.
#ifndef SF_B3_DEEP
#define SF_B3_DEEP 0
#endif
template <int MT, int NT, int WM, int WN>
constexpr int b3_nb() {
  if (!SF_B3_DEEP || WM * WN < 8) return 2;      // the 4-wave tiles keep two buffers and several workgroups per CU
  const int per = 16 * (MT * WM + NT * WN) * 128, n = (150 * 1024) / per;
  return n < 2 ? 2 : (n > 6 ? 6 : n);
}
template <int MT, int NT, int WM, int WN, int EPI, int NB, bool SCALE, bool B3 = false>
__global__ __launch_bounds__(64 * WM * WN, (B3 && !SF_B3_DEEP && WM * WN >= 8 ? 4 : 1)) void conv_glds_kernel(const ConvLaunch L) {
  constexpr int NWV = WM * WN;
  constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  constexpr int ROWS = BM + BN;
  static_assert((BM / 8) % NWV == 0 && (BN / 8) % NWV == 0, "A and B rows must split evenly over the waves");
  constexpr int GA = BM / 8 / NWV, GB = BN / 8 / NWV;   // LDS-DMA instructions per wave per chunk: weights / pixels
  constexpr int G = GA + GB;
  constexpr int LA = NB - 1;                     // chunks in flight
  constexpr int BUF = ROWS * 32;                 // floats per buffer
  // XOR mask of the 16-byte slot swizzle (row >> 1) & SWM.  fp32 loop: ds_read_b64 fragments, lane groups {0-31}, {32-63} -> 7.
  // bf16x3 loop: ds_read_b128 of slots 2g / 2g+1; its lane groups are {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32)
  // (MI355X_MICROARCH.md, LDS table): with mask 7 rows j and j+4.. of the g / g+1 halves met on one slot (measured
  // SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE); mask 5 gives the 16 lanes of every group 16 different slots
  constexpr int SWM = B3 ? 5 : 7;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [NB][ROWS][32]

  SF_STAMP_AT(L, 0);
  const ConvProblem& P = L.p[blockIdx.y];
  const int Ptot = P.n_img * P.Hout * P.Wout;
  const int n_mt = (P.cout_pad + BM - 1) / BM;
  // XCD-aware tile order for large launches.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
  // workgroup b runs on XCD b % 8.  With the plain order the three pixel rows a 3x3 tile reads are tiles of three
  // different XCDs and every L2 fetches them again (6.1 GB of fabric traffic per launch of the dominant kernel for
  // 3.2 GB of algorithmic bytes); here XCD x takes the chunks x, x+8, ... of 2^k consecutive tiles, which keeps the
  // halo rows in one L2 (3.8 GB) while the chunks still balance the tail.  The host pads gridDim.x to 8 chunks and
  // only asks for it when a launch has thousands of tiles (a small launch would leave XCDs idle).
  int bid = (int)blockIdx.x;
  if (L.xcd_shift) {      // block-uniform
    const int sh = L.xcd_shift - 1, i = bid >> 3;
    bid = ((((i >> sh) << 3) + (bid & 7)) << sh) + (i & ((1 << sh) - 1));
  }
  const int m_tile = bid % n_mt;
  const int p_tile = bid / n_mt;
  if (p_tile * BN >= Ptot) return;   // block-uniform

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane & 15, g = lane >> 4;

  const int img0 = (p_tile * BN) / (P.Hout * P.Wout);          // block-uniform
  const size_t img0_px = (size_t)img0 * P.Hin * P.Win;
  const float* const in0 = P.in0 + img0_px * P.in0_cs;
  const float* const in1 = P.in1 ? P.in1 + img0_px * P.in1_cs : nullptr;
  const int c0 = P.c0, c01 = P.c0 + P.c1;
  const int in0_cs = P.in0_cs, in1_cs = P.in1_cs;
  const int Win = P.Win, in_up = P.in_up, dil = P.dil, KW = P.KW;
  const int Hlog = P.Hin << P.in_up, Wlog = P.Win << P.in_up;
  const int HWout = P.Hout * P.Wout;
  const int* const gather = P.gather;
  const int KHg = P.KH;
  // SE-scaled input (res_models.py:161-165 feeding the next conv): a DMA cannot multiply, so the per-(image, channel)
  // scale is applied to the pixel fragments after they are read from LDS.  The scale rows of the (at most SC_IMGS)
  // images this tile touches sit behind the staging buffers; a lane's pixel of n-tile n belongs to image slot simg[n].
  constexpr int SC_IMGS = 4;
  const float* const in_scale = SCALE ? P.in_scale : nullptr;       // block-uniform; the plain instantiation carries none of this
  float* const sc_lds = smem + NB * BUF;          // [SC_IMGS][cin_pad]
  const int cin_pad = P.cin_pad;
  int simg[NT];
  if (SCALE && in_scale) {
    for (int idx = tid; idx < SC_IMGS * cin_pad; idx += 64 * NWV) {
      const int si = idx / cin_pad, c = idx - si * cin_pad;
      sc_lds[idx] = (img0 + si < P.n_img && c < c0) ? in_scale[(size_t)(img0 + si) * c0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int gp = p_tile * BN + (wn * NT + n) * 16 + j;
      simg[n] = (gp < Ptot ? gp / HWout - img0 : 0) * cin_pad;
    }
  }

  // staging slots.  Weight slot q (q < GA) of this wave fills rows (wave*GA + q)*8 .. +7 of the A block, pixel slot q
  // rows (wave*GB + q)*8 .. +7 of the B block; this lane: row + lane/8, 16-B slot lane%8 holding K values
  // 4*k4 .. 4*k4+3 with k4 = slot ^ ((row >> 1) & 7)   (row counted inside the buffer: A rows first)
  int a_voff[GA];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int r = (wave * GA + q) * 8 + (lane >> 3);
    const int k4 = (lane & 7) ^ ((r >> 1) & SWM);
    int grow = m_tile * BM + r;
    grow = grow < P.cout_pad ? grow : P.cout_pad - 1;
    a_voff[q] = (grow * P.ktot + k4 * 4) * (int)sizeof(float);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  auto make_rsrc = [](const float* base, size_t bytes) {
    const unsigned nrec = bytes < 0x7fffffffull ? (unsigned)bytes : 0x7fffffffu;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (int)nrec, 0x00020000);
  };
  const size_t imgs_left = (size_t)(P.n_img - img0);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(B3 ? static_cast<const float*>(P.w3) : P.w, (size_t)P.cout_pad * P.ktot * sizeof(float));
  const __amdgpu_buffer_rsrc_t rsrc0 = make_rsrc(in0, imgs_left * P.Hin * P.Win * in0_cs * sizeof(float));
  const __amdgpu_buffer_rsrc_t rsrc1 = make_rsrc(in1 ? in1 : in0, in1 ? imgs_left * P.Hin * P.Win * in1_cs * sizeof(float) : 0);
#endif
  int b_c4[GB], b_iy0[GB], b_ix0[GB], b_base[GB];
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int pr = (wave * GB + q) * 8 + (lane >> 3);
    b_c4[q] = 4 * ((lane & 7) ^ (((BM + pr) >> 1) & SWM));
    const int gp = p_tile * BN + pr;
    const bool v = gp < Ptot;
    const int img = v ? gp / HWout : 0;
    const int rem = gp - img * HWout;
    const int oy = rem / P.Wout, ox = rem - oy * P.Wout;
    b_iy0[q] = v ? oy * P.stride - P.pad : -(1 << 28);
    b_ix0[q] = ox * P.stride - P.pad;
    b_base[q] = (v ? img - img0 : 0) * P.Hin * P.Win;
  }

  const int kcpt = P.cin_pad / BK;
  const int nchunks_all = P.KH * P.KW * kcpt;
  // cross-workgroup split-K: this workgroup owns chunks [cb, cb + nchunks)
  const int nsplit = P.nsplit > 1 ? P.nsplit : 1;
  if ((int)blockIdx.z >= nsplit) return;   // block-uniform
  const int cps = (nchunks_all + nsplit - 1) / nsplit;
  const int cb = (int)blockIdx.z * cps;
  // sparse convolution with a tap mask (ConvProblem::tap_mask): the taps that are live for this tile's rows, 0 = walk every tap
  unsigned live_taps = 0;
  if (gather && P.tap_mask && nsplit == 1 && KW == 1 && KHg <= 32 && (BN % 64) == 0) {      // block-uniform
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < BN / 64; ++i) {
      const int idx = p_tile * (BN / 64) + i;
      if (idx < P.tap_mask_n) m |= P.tap_mask[idx];
    }
    if (KHg < 32) m &= (1u << KHg) - 1u;
    live_taps = m;      // (no live tap at all cannot happen for a stored site; 0 falls back to the full walk)
  }
  const int nchunks = live_taps ? __popc(live_taps) * kcpt : ((nchunks_all - cb) < cps ? (nchunks_all - cb) : cps);

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int cur_kc = cb % kcpt, cur_ty = live_taps ? __ffs((int)live_taps) - 1 : (cb / kcpt) / KW, cur_tx = (cb / kcpt) % KW;     // cursor of the next chunk to ISSUE
  // per tap and pixel slot: byte offset of the gathered pixel in either source, this lane's 16-byte channel slot included, or
  // 0x80000000 outside the image / the table (beyond any buffer: the range check zero-fills).  Per chunk a DMA then needs no
  // vector arithmetic at all — the channel chunk goes into the scalar offset (fp32 MFMAs share the vector ALU)
  int tap_off0[GB], tap_off1[GB];
  bool tap_fresh = true;
  const bool whole_chunks = (c01 % BK) == 0;      // block-uniform: no channel padding inside a chunk

  typedef __attribute__((address_space(3))) void lds_void;
  // one LDS-DMA of the chunk at the cursor: slot q of G (weights first); the last slot advances the cursor
  auto issue_one = [&](int chunk, int buf, int q) {
#if SF_GDIAG == 3   // probe: every chunk re-reads the addresses of chunk 0
    chunk = 0;
#endif
#if SF_GDIAG == 4
    return;
#endif
#if SF_GDIAG == 2
    if (q < GA) return;
#endif
#if SF_GDIAG == 1
    if (q >= GA) return;
#endif
    if (q < GA) {
      float* dst = smem + buf * BUF + (wave * GA + q) * 8 * 32;
#if defined(__HIP_DEVICE_COMPILE__)     // the host pass must not see the target builtins (it silently drops the kernel stub)
      const int wchunk = live_taps ? cur_ty * kcpt + cur_kc : chunk;      // (the cursor still points at this chunk: it moves with the last slot)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)dst, 16, a_voff[q], wchunk * (BK * 4), 0, 0);
#else
      (void)dst; (void)chunk;
#endif
    } else {
      const int qb = q - GA;
      if (tap_fresh && qb == 0) {
        if (gather) {      // block-uniform: sparse convolution, the neighbour table replaces the pixel arithmetic
#pragma unroll
          for (int i = 0; i < GB; ++i) {
            const int gp = p_tile * BN + (wave * GB + i) * 8 + (lane >> 3);
            const int px = gp < Ptot ? gather[(size_t)gp * KHg + cur_ty] : -1;
            tap_off0[i] = px >= 0 ? (px * in0_cs + b_c4[i]) * 4 : (int)0x80000000;
            tap_off1[i] = px >= 0 ? (px * in1_cs + b_c4[i]) * 4 : (int)0x80000000;
          }
        } else {
#pragma unroll
          for (int i = 0; i < GB; ++i) {
            const int iy = b_iy0[i] + cur_ty * dil, ix = b_ix0[i] + cur_tx * dil;
            const bool in = (iy >= 0) & (iy < Hlog) & (ix >= 0) & (ix < Wlog);
            const int px = in ? b_base[i] + (iy >> in_up) * Win + (ix >> in_up) : 0;
            tap_off0[i] = in ? (px * in0_cs + b_c4[i]) * 4 : (int)0x80000000;
            tap_off1[i] = in ? (px * in1_cs + b_c4[i]) * 4 : (int)0x80000000;
          }
        }
      }
      float* dst = smem + buf * BUF + (BM + (wave * GB + qb) * 8) * 32;
#if defined(__HIP_DEVICE_COMPILE__)
      const bool from1 = cur_kc * BK >= c0;                               // wave-uniform: the whole chunk reads in1
      int voff = from1 ? tap_off1[qb] : tap_off0[qb];
      if (!whole_chunks) voff = (cur_kc * BK + b_c4[qb] < c01) ? voff : (int)0x80000000;      // channels past cin inside the chunk: zero
      if (from1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (lds_void*)dst, 16, voff, (cur_kc * BK - c0) * 4, 0, 0);      // the vector offset alone is range-checked: it must not go negative
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (lds_void*)dst, 16, voff, cur_kc * (BK * 4), 0, 0);
#else
      (void)dst;
#endif
    }
#if SF_GDIAG != 3
    if (q == G - 1) {
      ++cur_kc;
      tap_fresh = false;
      if (cur_kc == kcpt) {
        cur_kc = 0;
        tap_fresh = true;
        if (live_taps) {      // the next live tap
          const unsigned rest = cur_ty + 1 < 32 ? live_taps >> (cur_ty + 1) : 0u;
          cur_ty = rest ? cur_ty + __ffs((int)rest) : KHg;
        } else if (++cur_tx == KW) { cur_tx = 0; ++cur_ty; }
      }
    }
#endif
  };

  const int sx = (j >> 1) & SWM;
  if constexpr (B3) {
    // ---- split-bf16 K loop: one v_mfma_f32_16x16x32_bf16 per (tile pair, product) and 32-deep chunk.  Lane (j, g) holds the
    // 8 K values 8g .. 8g+7 of its row: the 16-byte slots 2g and 2g+1 (weights: hi and lo pieces; pixels: two float4).
    // No register double-buffering and <= 100 VGPRs: TWO workgroups share a CU and one's barrier / DMA wait is the other's MFMA
    // time (measured against a software-pipelined loop with 2 / 3 / 4 staging buffers: profiles/README.md round-3 log).
    const int o0 = 4 * ((2 * g) ^ sx), o1 = 4 * ((2 * g + 1) ^ sx);
    f32x4 wa[1][MT][2], xb[1][NT][2], xs[1][NT][2];
    int kc_cmp = cb % kcpt;
    auto read3 = [&](int buf, auto SET) {
      constexpr int st = decltype(SET)::value;
      const float* a = smem + buf * BUF + (wm * MT * 16 + j) * 32;
      const float* b = smem + buf * BUF + (BM + wn * NT * 16 + j) * 32;
#pragma unroll
      for (int m = 0; m < MT; ++m) { wa[st][m][0] = lds_read_b128(a + m * 16 * 32 + o0); wa[st][m][1] = lds_read_b128(a + m * 16 * 32 + o1); }
#pragma unroll
      for (int n = 0; n < NT; ++n) { xb[st][n][0] = lds_read_b128(b + n * 16 * 32 + o0); xb[st][n][1] = lds_read_b128(b + n * 16 * 32 + o1); }
      if (SCALE && in_scale) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          xs[st][n][0] = lds_read_b128(sc_lds + simg[n] + kc_cmp * BK + 8 * g);
          xs[st][n][1] = lds_read_b128(sc_lds + simg[n] + kc_cmp * BK + 8 * g + 4);
        }
      }
    };
    {
      SF_STAMP_AT(L, 1);
#pragma unroll
      for (int c = 0; c < LA; ++c)
        if (c < nchunks) {      // block-uniform
#pragma unroll
          for (int q = 0; q < G; ++q) issue_one(cb + c, c, q);
        }
      SF_STAMP_AT(L, 2);
      int bufc = 0, ibuf = LA % NB;
      for (int c = 0; c < nchunks; ++c) {
        // chunk c landed (this wave's pieces; the LA - 1 younger chunks stay in flight) ...
        if (LA > 1 && c + LA - 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (LA - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // ... everybody's; and every wave is done with the buffer of chunk c - 1
        if (c + LA < nchunks) {
#pragma unroll
          for (int q = 0; q < G; ++q) issue_one(cb + c + LA, ibuf, q);
        }
        read3(bufc, std::integral_constant<int, 0>());
        bf16x8 bh[NT], bl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          f32x4 x0 = xb[0][n][0], x1 = xb[0][n][1];
          if (SCALE && in_scale) { x0 = x0 * xs[0][n][0]; x1 = x1 * xs[0][n][1]; }
          split_bf16x8(x0, x1, bh[n], bl[n]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[0][m][1]), bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[0][m][0]), bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[0][m][0]), bh[n], acc[m][n], 0, 0, 0);
          }
        kc_cmp = kc_cmp + 1 == kcpt ? 0 : kc_cmp + 1;
        bufc = bufc == NB - 1 ? 0 : bufc + 1;
        ibuf = ibuf == NB - 1 ? 0 : ibuf + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      SF_STAMP_AT(L, 3);
      if (nsplit > 1) {      // block-uniform
        __syncthreads();
        if (!splitk_handoff<MT, NT, NWV>(P, acc, nsplit, bid, wave, lane, tid, smem)) return;
      }
      SF_STAMP_AT(L, 4);
      run_epilogue<MT, NT, EPI>(P, acc, m_tile * BM + wm * MT * 16, p_tile * BN + wn * NT * 16, lane, Ptot, HWout);
      return;
    }
  }
  auto koff = [&](int t4) { return 4 * ((2 * t4 + (g >> 1)) ^ sx) + ((2 * g) & 3); };
  // Barrier in the middle of the MFMA stream (two buffers): the last k-group of chunk c is multiplied AFTER the
  // barrier that publishes chunk c+1, from fragments read before it, and the first fragments of chunk c+1 are read
  // under those MFMAs — a wave leaves the barrier with 2*MT*NT MFMAs ready to issue instead of a DMA-issue +
  // LDS-read-latency bubble.  The DMAs of chunk c+2 go out right after the same barrier (its buffer is free then).
  constexpr int NG = BK / 8;
  float2 fa[2][MT], fb[2][NT], fs[2][NT];
  int kc_cmp = cb % kcpt;     // channel chunk of the chunk being multiplied (the issue cursor runs ahead)
  auto read_frags = [&](int buf, int t4, int set) {
    const float* a = smem + buf * BUF + (wm * MT * 16 + j) * 32;
    const float* b = smem + buf * BUF + (BM + wn * NT * 16 + j) * 32;
#pragma unroll
    for (int m = 0; m < MT; ++m) fa[set][m] = lds_read_b64(a + m * 16 * 32 + koff(t4));
#pragma unroll
    for (int n = 0; n < NT; ++n) fb[set][n] = lds_read_b64(b + n * 16 * 32 + koff(t4));
    if (SCALE && in_scale) {      // logical K pair of this lane in the k-group: 8*t4 + 2*g, +1 (the slot swizzle only moves storage)
#pragma unroll
      for (int n = 0; n < NT; ++n) fs[set][n] = lds_read_b64(sc_lds + simg[n] + kc_cmp * BK + 8 * t4 + 2 * g);
    }
  };
  auto mfmas = [&](int set) {
    if (SCALE && in_scale) {
#pragma unroll
      for (int n = 0; n < NT; ++n) { fb[set][n].x *= fs[set][n].x; fb[set][n].y *= fs[set][n].y; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][m].x, fb[set][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][m].y, fb[set][n].y, acc[m][n], 0, 0, 0);
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // prologue: LA = NB-1 chunks in flight, the first one retired and published
  SF_STAMP_AT(L, 1);
#pragma unroll
  for (int c = 0; c < LA; ++c)
    if (c < nchunks) {      // block-uniform
#pragma unroll
      for (int q = 0; q < G; ++q) issue_one(cb + c, c, q);
    }
  if (LA > 1 && nchunks >= LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (LA - 1)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  SF_STAMP_AT(L, 2);
  // Iteration c starts right after the barrier that published chunk c: read its first fragments, multiply the LAST
  // k-group of chunk c-1 (read before the barrier) under that latency, start the DMAs of chunk c+LA into the buffer
  // the barrier just freed (chunk c-1's), then k-groups 0..NG-2 of chunk c, then retire chunk c+1 (a counted vmcnt:
  // the LA-1 younger chunks stay in flight).  No LDS read is in flight across the back edge, so the compiler's
  // lgkmcnt bookkeeping stays exact.
  // The two staging buffers are compile-time constants of the two halves of an unrolled pair of chunks: the fragment
  // addresses are then loop-invariant registers + an immediate ds_read offset (buffer, tile row) instead of 8 v_add per chunk
  // — fp32 MFMAs share the vector ALU, every VALU instruction in this loop is matrix time.
  static_assert(B3 || (NB == 2 && LA == 1), "two staging buffers");
#ifdef SF_STAMP
  unsigned long long cyc_issue = 0, cyc_wait = 0, cyc_bar = 0;
  const unsigned long long cyc_loop0 = __builtin_amdgcn_s_memtime();
#endif
  auto chunk_step = [&](const int c, auto BUFC) {
    constexpr int buf = decltype(BUFC)::value, ibuf = buf ^ 1;
    const bool more = c + LA < nchunks;
    read_frags(buf, 0, 0);
    if (c > 0) mfmas((NG - 1) & 1);
#ifdef SF_STAMP
    const unsigned long long ti0 = __builtin_amdgcn_s_memtime();
#endif
    if (more) {
#pragma unroll
      for (int q = 0; q < G; ++q) issue_one(cb + c + LA, ibuf, q);
    }
#ifdef SF_STAMP
    cyc_issue += __builtin_amdgcn_s_memtime() - ti0;
#endif
#pragma unroll
    for (int t4 = 0; t4 < NG - 1; ++t4) {
      read_frags(buf, t4 + 1, (t4 + 1) & 1);
      mfmas(t4 & 1);
    }
    if (c + 1 < nchunks) {
#ifdef SF_STAMP
      const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
      if (LA > 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (LA - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SF_STAMP
      const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
      cyc_wait += tw1 - tw0;
#endif
      __builtin_amdgcn_s_barrier();
#ifdef SF_STAMP
      cyc_bar += __builtin_amdgcn_s_memtime() - tw1;
#endif
    }
    kc_cmp = kc_cmp + 1 == kcpt ? 0 : kc_cmp + 1;
  };
  for (int c = 0; c < nchunks; c += 2) {
    chunk_step(c, std::integral_constant<int, 0>());
    if (c + 1 < nchunks) chunk_step(c + 1, std::integral_constant<int, 1>());
  }
  mfmas((NG - 1) & 1);
  SF_STAMP_AT(L, 3);
#ifdef SF_STAMP
  SF_STAMP_VAL(L, 8, __builtin_amdgcn_s_memtime() - cyc_loop0);
  SF_STAMP_VAL(L, 9, cyc_issue);
  SF_STAMP_VAL(L, 10, cyc_wait);
  SF_STAMP_VAL(L, 11, cyc_bar);
  SF_STAMP_VAL(L, 12, (unsigned long long)nchunks);
#endif
  if (nsplit > 1) {      // block-uniform
    __syncthreads();     // the hand-off flag lives in the staging buffers: every wave is done reading them
    if (!splitk_handoff<MT, NT, NWV>(P, acc, nsplit, bid, wave, lane, tid, smem)) return;
  }
  SF_STAMP_AT(L, 4);
  run_epilogue<MT, NT, EPI>(P, acc, m_tile * BM + wm * MT * 16, p_tile * BN + wn * NT * 16, lane, Ptot, HWout);
#ifdef SF_STAMP
  SF_STAMP_AT(L, 5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SF_STAMP_AT(L, 6);
#endif
}

template <int MT, int NT, int WM, int WN, int EPI, bool SCALE, bool B3>
static hipError_t launch_glds_tb(const ConvLaunch& L, hipStream_t stream);
// bf16x3 (opt-in math mode): every problem of the launch carries split weights (api.hip decides); block-uniform host switch
template <int MT, int NT, int WM, int WN, int EPI, bool SCALE = false>
static hipError_t launch_glds_t(const ConvLaunch& L, hipStream_t stream) {
  bool b3 = L.nprob > 0;
  for (int i = 0; i < L.nprob; ++i) b3 = b3 && L.p[i].w3 != nullptr && L.p[i].use_w3;
  return b3 ? launch_glds_tb<MT, NT, WM, WN, EPI, SCALE, true>(L, stream) : launch_glds_tb<MT, NT, WM, WN, EPI, SCALE, false>(L, stream);
}
template <int MT, int NT, int WM, int WN, int EPI, bool SCALE, bool B3>
static hipError_t launch_glds_tb(const ConvLaunch& L, hipStream_t stream) {
  constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  constexpr int NB = B3 ? b3_nb<MT, NT, WM, WN>() : 2;
  constexpr int lds = NB * (BM + BN) * 32 * 4 + (SCALE ? 4 * 256 * 4 : 0);   // staging buffers (+ the SE scale rows of up to 4 images x 256 channels)
  auto kern = conv_glds_kernel<MT, NT, WM, WN, EPI, NB, SCALE, B3>;
  // the attribute is per device (a process may touch more than one GPU); one process per GPU and one launching
  // thread per device are the documented use (include/sfnative.h), so the flag needs no lock
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  int maxblocks = 0;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProblem& P = L.p[i];
    int Ptot = P.n_img * P.Hout * P.Wout;
    int nb = ((Ptot + BN - 1) / BN) * ((P.cout_pad + BM - 1) / BM);
    if (nb > maxblocks) maxblocks = nb;
  }
  if (maxblocks == 0) return hipSuccess;
  int zs = 1;
  for (int i = 0; i < L.nprob; ++i) zs = L.p[i].nsplit > zs ? L.p[i].nsplit : zs;
  ConvLaunch LL = L;
  LL.xcd_shift = 0;
  if (xcd_chunk_log2() >= 0 && maxblocks >= 4096 && zs == 1) {      // see the tile order in the kernel
    const int span = 8 << xcd_chunk_log2();
    maxblocks = (maxblocks + span - 1) / span * span;
    LL.xcd_shift = xcd_chunk_log2() + 1;
  }
  hipLaunchKernelGGL(kern, dim3(maxblocks, L.nprob, zs), dim3(64 * WM * WN), lds, stream, LL);
  return hipGetLastError();
}

template <int MT, int NT, int WM, int WN>
static hipError_t launch_glds_e(const ConvLaunch& L, int epi, hipStream_t stream) {
  if (epi == EPI_AFFINE) return launch_glds_t<MT, NT, WM, WN, EPI_AFFINE>(L, stream);
  if (epi == EPI_BLEND) return launch_glds_t<MT, NT, WM, WN, EPI_BLEND>(L, stream);
  return hipErrorInvalidValue;
}
// SE-scaled inputs (the two p_model layers behind an SELayer): affine and sampling epilogues
template <int MT, int NT, int WM, int WN>
static hipError_t launch_glds_s(const ConvLaunch& L, int epi, hipStream_t stream) {
  if (epi == EPI_AFFINE) return launch_glds_t<MT, NT, WM, WN, EPI_AFFINE, true>(L, stream);
  if (epi == EPI_SAMPLE) return launch_glds_t<MT, NT, WM, WN, EPI_SAMPLE, true>(L, stream);
  return hipErrorInvalidValue;
}

// diagnostic: workgroups per CU the runtime grants the 128 x 128 and 64 x 128 (8-wave) AFFINE tiles with their LDS (tools/r03/occupancy.py)
int glds_occupancy(int which) {
  int n = -1;
  hipError_t e = hipErrorInvalidValue;
  if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_glds_kernel<4, 2, 2, 4, EPI_AFFINE, 2, false, false>, 512, 2 * 256 * 32 * 4);
  if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_glds_kernel<4, 2, 2, 4, EPI_AFFINE, b3_nb<4, 2, 2, 4>(), false, true>, 512, b3_nb<4, 2, 2, 4>() * 256 * 32 * 4);
  if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_glds_kernel<2, 2, 2, 4, EPI_AFFINE, 2, false, false>, 512, 2 * 192 * 32 * 4);
  if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_glds_kernel<2, 2, 2, 4, EPI_AFFINE, b3_nb<2, 2, 2, 4>(), false, true>, 512, b3_nb<2, 2, 2, 4>() * 192 * 32 * 4);
  return e == hipSuccess ? n : -1;
}

// tile: 0 = 128 cout x 128 px (2x4 waves of 64x32), 1 = 64 x 64 (2x2 waves of 32x32);  variant: buffers / issue placement
hipError_t launch_conv_glds(const ConvLaunch& L, int epi, int tile, int variant, hipStream_t stream) {
  bool scaled = false;
  for (int i = 0; i < L.nprob; ++i) scaled = scaled || (L.p[i].in_scale != nullptr);
  if (scaled || epi == EPI_SAMPLE) {     // 64x64 tiles (large P), 32x32 (one latent)
    if (tile == 3) return launch_glds_s<1, 1, 2, 2>(L, epi, stream);
    if (tile == 1 || tile == 0) return launch_glds_s<2, 2, 2, 2>(L, epi, stream);
    if (tile == 4) return launch_glds_s<4, 1, 1, 4>(L, epi, stream);      // cross-workgroup split-K launches (a few batched samples)
    return hipErrorInvalidValue;
  }
  if (tile == 2) {   // LayerNorm epilogues: one wave holds all (<= 64) output channels of its pixels; 64 cout x 128 px, 4 waves
    if (epi == EPI_LNG) return launch_glds_t<4, 2, 1, 4, EPI_LNG>(L, stream);
    if (epi == EPI_TRUST) return launch_glds_t<4, 2, 1, 4, EPI_TRUST>(L, stream);
    return hipErrorInvalidValue;
  }
  if (tile == 5) {   // LayerNorm epilogues over 65..128 output channels (hidden sizes no shipped config uses): 128 cout x 64 px, 4 waves of 128x16
    if (epi == EPI_LNG) return launch_glds_t<8, 1, 1, 4, EPI_LNG>(L, stream);
    if (epi == EPI_TRUST) return launch_glds_t<8, 1, 1, 4, EPI_TRUST>(L, stream);
    return hipErrorInvalidValue;
  }
  if (tile == 4) {   // cross-workgroup split-K launches: 64 cout x 64 px, 4 waves of 64x16 (all channels of a pixel in one wave)
    switch (epi) {
      case EPI_AFFINE: return launch_glds_t<4, 1, 1, 4, EPI_AFFINE>(L, stream);
      case EPI_BLEND:  return launch_glds_t<4, 1, 1, 4, EPI_BLEND>(L, stream);
      case EPI_LNG:    return launch_glds_t<4, 1, 1, 4, EPI_LNG>(L, stream);
      case EPI_TRUST:  return launch_glds_t<4, 1, 1, 4, EPI_TRUST>(L, stream);
    }
    return hipErrorInvalidValue;
  }
  if (tile == 3) {   // small pixel counts (one 50x50 latent): 32 cout x 32 px, 4 waves of 16x16
    return launch_glds_e<1, 1, 2, 2>(L, epi, stream);
  }
  if (tile == 0) return launch_glds_e<4, 2, 2, 4>(L, epi, stream);     // 128 cout x 128 px, 8 waves of 64x32
  switch (variant) {
    case 4: return launch_glds_e<2, 2, 2, 2>(L, epi, stream);   // 64 cout x 64 px, 4 waves of 32x32
    case 6: return launch_glds_e<2, 2, 2, 4>(L, epi, stream);   // 64 cout x 128 px, 8 waves of 32x32
    case 7: return launch_glds_e<4, 2, 1, 4>(L, epi, stream);   // 64 cout x 128 px, 4 waves of 64x32
    case 9: return launch_glds_e<1, 4, 2, 2>(L, epi, stream);   // 32 cout x 128 px, 4 waves of 16x64 (narrow layers: 16- / 32-channel sparse stages)
    case 10: return launch_glds_e<2, 4, 2, 4>(L, epi, stream);  // 64 cout x 256 px, 8 waves of 32x64 (64-channel layers at large P: the staged bytes per FLOP of the 128 x 128 tile)
  }
  return hipErrorInvalidValue;
}

// ---- direct-fragment kernel (small pixel counts: the 50x50 BEV latent of the GRU-ODE) -------
// At 2500 pixels a layer has only 157 pixel tiles: nothing is shared between the waves of a
// workgroup (each K-group has its own K slice), so staging through LDS buys no reuse and only
// adds a write+read+barrier to every chunk while 92 KB of LDS pin occupancy to one workgroup per CU.
// Here every wave loads its MFMA fragments straight from global memory (L2-resident): within a
// 32-deep chunk lane (j, g) owns the 8 consecutive K values 8g..8g+7 of weight row j / pixel j
// (two 16-B loads, the four g-lanes of a row cover one full 128-B line), and MFMA step e
// contracts the K slice {8g+e}.  No LDS and no barrier in the K loop, waves run free with one
// chunk of register prefetch; occupancy is bounded by VGPRs only.  blockDim.x/64 = number of
// K-groups (in-workgroup split-K, fixed-order LDS reduction at the end).
template <int MT, int EPI>
__global__ __launch_bounds__(512) void conv_direct_kernel(const ConvLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  SF_STAMP_AT(L, 0);
  const ConvProblem& P = L.p[blockIdx.y];
  const int Ptot = P.n_img * P.Hout * P.Wout;
  constexpr int BM = 16 * MT;
  const int n_mt = (P.cout_pad + BM - 1) / BM;
  const int m_tile = blockIdx.x % n_mt;
  const int p_tile = blockIdx.x / n_mt;
  if (p_tile * 16 >= Ptot) return;   // block-uniform

  const int KS = blockDim.x >> 6;
  const int kg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;

  const float* const in0 = P.in0;
  const float* const in1 = P.in1;
  const float* const gate = P.gate;
  const float* const in_scale = P.in_scale;
  const int c0 = P.c0, c01 = P.c0 + P.c1;
  const int in0_cs = P.in0_cs, in1_cs = P.in1_cs, gate_cs = P.gate_cs, gate_co = P.gate_co;
  const int Win = P.Win, in_up = P.in_up, dil = P.dil, KW = P.KW;
  const int Hlog = P.Hin << P.in_up, Wlog = P.Win << P.in_up;
  const bool has_aux = (gate != nullptr) | (in_scale != nullptr);
  const int HWout = P.Hout * P.Wout;

  // this lane's pixel
  const int gp = p_tile * 16 + j;
  const bool pvalid = gp < Ptot;
  const int img = pvalid ? gp / HWout : 0;
  const int rem = gp - img * HWout;
  const int oy = rem / P.Wout, ox = rem - oy * P.Wout;
  const int iy0 = pvalid ? oy * P.stride - P.pad : -(1 << 28);
  const int ix0 = ox * P.stride - P.pad;
  const size_t pbase = (size_t)img * P.Hin * P.Win;
  const float* const sc_row = in_scale ? in_scale + (size_t)img * c0 : in0;

  // this lane's weight rows (clamped: rows >= cout_pad are never stored)
  const float* a_ptr[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int row = m_tile * BM + m * 16 + j;
    row = row < P.cout_pad ? row : P.cout_pad - 1;
    a_ptr[m] = P.w + (size_t)row * P.ktot + 8 * g;
  }

  const int kcpt = P.cin_pad / BK;
  const int nchunks = P.KH * P.KW * kcpt;
  const int n_my = kg < nchunks ? (nchunks - kg + KS - 1) / KS : 0;
  int cur_kc = kg % kcpt;
  int cur_ty = (kg / kcpt) / KW, cur_tx = (kg / kcpt) - cur_ty * KW;

  f32x4 acc[MT][1];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 a0[MT][2], a1[MT][2], b0[2], b1[2], x0[2], x1[2];
  int f0 = 0, f1 = 0;

  auto load = [&](float4 (&a)[MT][2], float4 (&b)[2], float4 (&x)[2], int& fl, int chunk) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      a[m][0] = ld4(a_ptr[m] + (size_t)chunk * BK);
      a[m][1] = ld4(a_ptr[m] + (size_t)chunk * BK + 4);
    }
    const int c = cur_kc * BK + 8 * g;
    const bool s0 = c < c0;
    const bool s1 = (!s0) & (c < c01);
    const int cc = c - c0;
    const int iy = iy0 + cur_ty * dil, ix = ix0 + cur_tx * dil;
    const bool ok = (iy >= 0) & (iy < Hlog) & (ix >= 0) & (ix < Wlog) & (s0 | s1);
    const size_t pix = ok ? pbase + (size_t)((iy >> in_up) * Win + (ix >> in_up)) : 0;
    const float* p = s1 ? in1 + pix * in1_cs + cc : in0 + pix * in0_cs + (s0 ? c : 0);
    b[0] = ld4(p);
    b[1] = ld4(p + 4);
    int f = ok ? 1 : 0;
    if (has_aux) {   // block-uniform
      const bool m1 = ok & s0 & (in_scale != nullptr);
      const bool m2 = ok & s1 & (gate != nullptr);
      const float* q = m1 ? sc_row + c : (m2 ? gate + pix * gate_cs + gate_co + cc : in0);
      x[0] = ld4(q);
      x[1] = ld4(q + 4);
      f |= (m1 ? 2 : 0) | (m2 ? 4 : 0);
    }
    fl = f;
    cur_kc += KS;
    while (cur_kc >= kcpt) {
      cur_kc -= kcpt;
      if (++cur_tx == KW) { cur_tx = 0; ++cur_ty; }
    }
  };

  auto compute = [&](float4 (&a)[MT][2], float4 (&b)[2], float4 (&x)[2], int fl) {
    float bv[8] = {b[0].x, b[0].y, b[0].z, b[0].w, b[1].x, b[1].y, b[1].z, b[1].w};
    if (has_aux) {
      const float xv[8] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w};
      const bool sc = fl & 2, gt = fl & 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] *= sc ? xv[e] : (gt ? 1.f - xv[e] : 1.f);
    }
    const bool ok = fl & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = ok ? bv[e] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 av = a[m][e >> 2];
        const float ae = (e & 3) == 0 ? av.x : ((e & 3) == 1 ? av.y : ((e & 3) == 2 ? av.z : av.w));
        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, bv[e], acc[m][0], 0, 0, 0);
      }
    }
  };

  SF_STAMP_AT(L, 1);
  if (n_my > 0) load(a0, b0, x0, f0, kg);
#ifdef SF_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SF_STAMP_AT(L, 2);
#endif
  for (int i = 0; i < n_my; i += 2) {
    if (i + 1 < n_my) load(a1, b1, x1, f1, kg + (i + 1) * KS);
    compute(a0, b0, x0, f0);
    if (i + 2 < n_my) load(a0, b0, x0, f0, kg + (i + 2) * KS);
    if (i + 1 < n_my) compute(a1, b1, x1, f1);
  }

  SF_STAMP_AT(L, 3);
  // fixed-order split-K reduction through LDS
  if (KS > 1) {
    constexpr int PER_WAVE = MT * 4 * 64;
    if (kg > 0) {
      float* r = smem + (kg - 1) * PER_WAVE + lane;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) r[(m * 4 + q) * 64] = acc[m][0][q];
    }
    __syncthreads();
    if (kg > 0) return;
    for (int s = 1; s < KS; ++s) {
      const float* r = smem + (s - 1) * PER_WAVE + lane;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][0][q] += r[(m * 4 + q) * 64];
    }
  }
  SF_STAMP_AT(L, 4);
  run_epilogue<MT, 1, EPI>(P, acc, m_tile * BM, p_tile * 16, lane, Ptot, HWout);
#ifdef SF_STAMP
  SF_STAMP_AT(L, 5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SF_STAMP_AT(L, 6);
#endif
}

template <int MT, int EPI>
static hipError_t launch_direct_t(const ConvLaunch& L, int ks, hipStream_t stream) {
  int maxblocks = 0;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProblem& P = L.p[i];
    int Ptot = P.n_img * P.Hout * P.Wout;
    int nb = ((Ptot + 15) / 16) * ((P.cout_pad + 16 * MT - 1) / (16 * MT));
    if (nb > maxblocks) maxblocks = nb;
  }
  if (maxblocks == 0) return hipSuccess;
  const int lds = (ks - 1) * MT * 4 * 64 * 4;
  hipLaunchKernelGGL((conv_direct_kernel<MT, EPI>), dim3(maxblocks, L.nprob), dim3(64 * ks), lds, stream, L);
  return hipGetLastError();
}

template <int EPI>
static hipError_t launch_direct_e(const ConvLaunch& L, int mt, int ks, hipStream_t stream) {
  switch (mt) {
    case 1: return launch_direct_t<1, EPI>(L, ks, stream);
    case 2: return launch_direct_t<2, EPI>(L, ks, stream);
    case 4: return launch_direct_t<4, EPI>(L, ks, stream);
  }
  return hipErrorInvalidValue;
}

// mt: 16-row cout tiles per wave (1, 2, 4); ks: K-groups per workgroup (1..8)
hipError_t launch_conv_direct(const ConvLaunch& L, int epi, int mt, int ks, hipStream_t stream) {
  if (ks < 1 || ks > 8) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_AFFINE: return launch_direct_e<EPI_AFFINE>(L, mt, ks, stream);
    case EPI_BLEND:  return launch_direct_e<EPI_BLEND>(L, mt, ks, stream);
    case EPI_LNG:    return launch_direct_e<EPI_LNG>(L, mt, ks, stream);
    case EPI_TRUST:  return launch_direct_e<EPI_TRUST>(L, mt, ks, stream);
    case EPI_SAMPLE: return launch_direct_e<EPI_SAMPLE>(L, mt, ks, stream);
  }
  return hipErrorInvalidValue;
}

// ---- host-side launcher --------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int KS, int EPI, int KB = 32, bool SWZ = false>
static hipError_t launch_cfg(const ConvLaunch& L, hipStream_t stream) {
  constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  constexpr int stage_bytes = KS * 2 * (BM + BN) * (SWZ ? KB : KB + 4) * 4;
  constexpr int red_bytes = (KS - 1) * WM * WN * MT * NT * 4 * 64 * 4;
  constexpr int lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  auto kern = conv_igemm_kernel<MT, NT, WM, WN, KS, EPI, KB, SWZ>;
  // the attribute is per device (a process may touch more than one GPU); one process per GPU and one launching
  // thread per device are the documented use (include/sfnative.h), so the flag needs no lock
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  int maxblocks = 0;
  for (int i = 0; i < L.nprob; ++i) {
    const ConvProblem& P = L.p[i];
    int Ptot = P.n_img * P.Hout * P.Wout;
    int nb = ((Ptot + BN - 1) / BN) * ((P.cout_pad + BM - 1) / BM);
    if (nb > maxblocks) maxblocks = nb;
  }
  if (maxblocks == 0) return hipSuccess;
  int zs = 1;
  for (int i = 0; i < L.nprob; ++i) zs = L.p[i].nsplit > zs ? L.p[i].nsplit : zs;
  dim3 grid(maxblocks, L.nprob, zs), block(64 * WM * WN * KS);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, L);
  return hipGetLastError();
}

// Tile configurations.  "S": small pixel counts (BEV latent 50x50): 16 px x 64 cout per
// workgroup, 4-way in-workgroup split-K.  "L": large pixel counts (200x200 head): 64x64 tile.
hipError_t launch_conv(const ConvLaunch& L, int epi, int cfg, hipStream_t stream) {
  switch (cfg) {
    case 0:   // S: MT4 NT1 1x1 waves, KS 4
      switch (epi) {
        case EPI_AFFINE: return launch_cfg<4, 1, 1, 1, 4, EPI_AFFINE>(L, stream);
        case EPI_BLEND:  return launch_cfg<4, 1, 1, 1, 4, EPI_BLEND>(L, stream);
        case EPI_LNG:    return launch_cfg<4, 1, 1, 1, 4, EPI_LNG>(L, stream);
        case EPI_TRUST:  return launch_cfg<4, 1, 1, 1, 4, EPI_TRUST>(L, stream);
        case EPI_SAMPLE: return launch_cfg<4, 1, 1, 1, 4, EPI_SAMPLE>(L, stream);
      }
      break;
    case 1:   // L: 2x2 waves of 32x32
      switch (epi) {
        case EPI_AFFINE: return launch_cfg<2, 2, 2, 2, 1, EPI_AFFINE>(L, stream);
        case EPI_BLEND:  return launch_cfg<2, 2, 2, 2, 1, EPI_BLEND>(L, stream);
        case EPI_SAMPLE: return launch_cfg<2, 2, 2, 2, 1, EPI_SAMPLE>(L, stream);
      }
      break;
    case 4:   // T: 4 waves x (64 cout x 16 px) = 64x64 tile, every epilogue; used with cross-WG split-K
      switch (epi) {
        case EPI_AFFINE: return launch_cfg<4, 1, 1, 4, 1, EPI_AFFINE>(L, stream);
        case EPI_BLEND:  return launch_cfg<4, 1, 1, 4, 1, EPI_BLEND>(L, stream);
        case EPI_LNG:    return launch_cfg<4, 1, 1, 4, 1, EPI_LNG>(L, stream);
        case EPI_TRUST:  return launch_cfg<4, 1, 1, 4, 1, EPI_TRUST>(L, stream);
        case EPI_SAMPLE: return launch_cfg<4, 1, 1, 4, 1, EPI_SAMPLE>(L, stream);
      }
      break;
    case 9:   // 128 x 128, 8 waves (2x4) of 64x32
      switch (epi) {
        case EPI_AFFINE: return launch_cfg<4, 2, 2, 4, 1, EPI_AFFINE>(L, stream);
        case EPI_BLEND:  return launch_cfg<4, 2, 2, 4, 1, EPI_BLEND>(L, stream);
      }
      break;
    case 2:   // LN-capable large tile: 4 waves x (64 cout x 32 px)
      switch (epi) {
        case EPI_LNG:    return launch_cfg<4, 2, 1, 4, 1, EPI_LNG>(L, stream);
        case EPI_TRUST:  return launch_cfg<4, 2, 1, 4, 1, EPI_TRUST>(L, stream);
      }
      break;
  }
  return hipErrorInvalidValue;
}

}  // namespace sf
