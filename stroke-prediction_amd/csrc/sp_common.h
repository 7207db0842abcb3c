// Shared device helpers for the gfx950 kernels of the U-Net / CAE hot path.
// CDNA4 only: wave = 64 lanes, MFMA 16x16x32 bf16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stroke_amd.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned short bf16_t;

#define SP_WAVE 64

// ---- error plumbing (thread-local message, see sp_last_error) ------------------------------
void sp_set_error(const char* fmt, ...);
#define SP_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      sp_set_error(__VA_ARGS__);           \
      return SP_EINVAL;                    \
    }                                      \
  } while (0)
#define SP_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      sp_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return SP_EHIP;                                                           \
    }                                                                           \
  } while (0)

// raise a kernel's dynamic-LDS limit once per (kernel, size) instead of on every launch
#define SP_ENSURE_LDS(kern, bytes, what)                                                                           \
  do {                                                                                                             \
    static int granted_ = 0;                                                                                       \
    if ((bytes) > 48 * 1024 && (bytes) > granted_) {                                                               \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
      if (e_ != hipSuccess) { sp_set_error("%s: cannot raise LDS limit to %d: %s", what, (int)(bytes), hipGetErrorString(e_)); return SP_EHIP; } \
      granted_ = (bytes);                                                                                          \
    }                                                                                                              \
  } while (0)

// ---- the 16-bit storage type ------------------------------------------------------------------
// bf16 (8 exponent / 7 mantissa bits) in libstroke_amd.so.  The SAME sources built with -DSP_HALF_F16 give
// libstroke_amd_f16.so, the "f16" precision mode: IEEE half (5 / 10 bits) -- three more mantissa bits at the same MFMA rate
// (v_mfma_f32_16x16x32_f16) and the same bytes; the range is what BatchNorm-ed activations need, and the engine scales the
// output gradients by a power of two (runtime/unet_engine.py) because 1e-7-sized gradients are below half's subnormals.
// Everything that depends on the format is in this block; the kernels keep the name bf16_t for "16-bit storage word".
#ifdef SP_HALF_F16
typedef _Float16 sp_h16;
#else
typedef __bf16 sp_h16;
#endif
typedef sp_h16 sp_h16x8 __attribute__((ext_vector_type(8)));
typedef sp_h16 sp_h16x4 __attribute__((ext_vector_type(4)));
typedef sp_h16 sp_bf16x2 __attribute__((ext_vector_type(2)));
// D = A B + C on 16-bit operands (any 16-byte / 8-byte vector types holding the storage words); the three modifier arguments
// of the builtin are always 0 here
template <typename A, typename B>
__device__ __forceinline__ f32x4 SP_MFMA16(A a, B b, f32x4 c, int = 0, int = 0, int = 0) {
#ifdef SP_HALF_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sp_h16x8, a), __builtin_bit_cast(sp_h16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sp_h16x8, a), __builtin_bit_cast(sp_h16x8, b), c, 0, 0, 0);
#endif
}
template <typename A, typename B>
__device__ __forceinline__ f32x4 SP_MFMA16_K16(A a, B b, f32x4 c, int = 0, int = 0, int = 0) {
#ifdef SP_HALF_F16
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(sp_h16x4, a), __builtin_bit_cast(sp_h16x4, b), c, 0, 0, 0);
#else
  typedef short sp_s16x4 __attribute__((ext_vector_type(4)));
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(sp_s16x4, a), __builtin_bit_cast(sp_s16x4, b), c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(sp_h16, h); }
// the low / high half of a dword holding two storage words
__device__ __forceinline__ float sp_h2f_lo(uint32_t w) {
#ifdef SP_HALF_F16
  return bf2f((bf16_t)(w & 0xffffu));
#else
  return __uint_as_float(w << 16);
#endif
}
__device__ __forceinline__ float sp_h2f_hi(uint32_t w) {
#ifdef SP_HALF_F16
  return bf2f((bf16_t)(w >> 16));
#else
  return __uint_as_float(w & 0xffff0000u);
#endif
}

// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
  sp_h16 b = (sp_h16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword holding two storage words (round to nearest even): ONE v_cvt_pk_bf16_f32.  Converting element by
// element and packing by hand costs a conversion, a shift and an or per element (found in the ISA of every bf16 epilogue).
__device__ __forceinline__ uint32_t sp_pack_bf16x2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, sp_bf16x2));
}

// storage-type traits: T = bf16_t (fast path) or float (parity path)
template <typename T> struct Store;
template <> struct Store<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  // 8 consecutive elements (16-byte aligned)
  static __device__ __forceinline__ void ld8(const bf16_t* p, float* v) {
    uint4 r = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = sp_h2f_lo(w[i]);
      v[2 * i + 1] = sp_h2f_hi(w[i]);
    }
  }
  static __device__ __forceinline__ void st8(bf16_t* p, const float* v) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = sp_pack_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ void st4(bf16_t* p, const float* v) {
    uint32_t w0 = sp_pack_bf16x2(v[0], v[1]);
    uint32_t w1 = sp_pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = make_uint2(w0, w1);
  }
  static __device__ __forceinline__ void ld4(const bf16_t* p, float* v) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    v[0] = sp_h2f_lo(r.x); v[1] = sp_h2f_hi(r.x);
    v[2] = sp_h2f_lo(r.y); v[3] = sp_h2f_hi(r.y);
  }
};
template <> struct Store<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void ld8(const float* p, float* v) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void st8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  static __device__ __forceinline__ void st4(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  static __device__ __forceinline__ void ld4(const float* p, float* v) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
};

// ---- bf16 PAIR tensors (SP_HL, the forward storage of the "bf16x3" precision mode): value = hi + lo, hi = bf16(value) in one
// tensor, lo = bf16(value - hi) in a second tensor of the same shape `lo_delta` bytes behind it (~17 significand bits).  The hi
// tensor alone IS the bf16 mode's tensor: the backward kernels read it unchanged.  sp_hl_t marks a pointer to the hi half.
struct sp_hl_t { bf16_t v; };
// four values -> their hi and lo words (two storage words per dword)
__device__ __forceinline__ void sp_hl_split4(const float* v, uint32_t& h0, uint32_t& h1, uint32_t& l0, uint32_t& l1) {
  h0 = sp_pack_bf16x2(v[0], v[1]);
  h1 = sp_pack_bf16x2(v[2], v[3]);
  l0 = sp_pack_bf16x2(v[0] - sp_h2f_lo(h0), v[1] - sp_h2f_hi(h0));
  l1 = sp_pack_bf16x2(v[2] - sp_h2f_lo(h1), v[3] - sp_h2f_hi(h1));
}
// eight consecutive channels of one voxel (16-byte aligned in both halves)
__device__ __forceinline__ void sp_hl_ld8(const sp_hl_t* p, int64_t lo_delta, float* v) {
  const uint4 h = *reinterpret_cast<const uint4*>(p);
  const uint4 l = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p) + lo_delta);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = sp_h2f_lo(hw[i]) + sp_h2f_lo(lw[i]);
    v[2 * i + 1] = sp_h2f_hi(hw[i]) + sp_h2f_hi(lw[i]);
  }
}
__device__ __forceinline__ void sp_hl_st8(sp_hl_t* p, int64_t lo_delta, const float* v) {
  uint32_t h[4], l[4];
  sp_hl_split4(v, h[0], h[1], l[0], l[1]);
  sp_hl_split4(v + 4, h[2], h[3], l[2], l[3]);
  *reinterpret_cast<uint4*>(p) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p) + lo_delta) = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void sp_hl_st4(sp_hl_t* p, int64_t lo_delta, const float* v) {
  uint32_t h0, h1, l0, l1;
  sp_hl_split4(v, h0, h1, l0, l1);
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p) + lo_delta) = make_uint2(l0, l1);
}
// eight channels of a voxel through one interface: T = bf16_t / float (lo_delta ignored) or sp_hl_t
template <typename T> __device__ __forceinline__ void ld8x(const T* p, int64_t, float* v) { Store<T>::ld8(p, v); }
template <> __device__ __forceinline__ void ld8x<sp_hl_t>(const sp_hl_t* p, int64_t lo_delta, float* v) { sp_hl_ld8(p, lo_delta, v); }
template <typename T> __device__ __forceinline__ void st8x(T* p, int64_t, const float* v) { Store<T>::st8(p, v); }
template <> __device__ __forceinline__ void st8x<sp_hl_t>(sp_hl_t* p, int64_t lo_delta, const float* v) { sp_hl_st8(p, lo_delta, v); }

// ---- optional fp8 shadow output of an elementwise kernel (the "fp8" precision mode, csrc/sp_conv_zm8.hip): besides its bf16
// tensor the kernel writes q8 = fp8(scale * bf16(value)) into a PLANE-MAJOR tensor [CP/16][nvox][16 bytes] -- the operand of
// the next fp8 convolution -- instead of leaving that to a separate pass over HBM (sp_quantize_f8).  p == nullptr: off.
struct SpQ8 {
  unsigned char* p;
  int64_t plane;      // bytes per 16-channel plane
  float scale;
  int32_t fmt;        // 0 = e4m3 (saturating at 448), 1 = e5m2 (57344)
};
__device__ __forceinline__ uint32_t sp_q8_pack4(const float* v, float s, int fmt) {
  float a[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = bf2f(f2bf(v[j])) * s;   // the stored (16-bit) value
  int r;
  if (fmt) {
    r = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(a[0], -57344.f, 57344.f), __builtin_amdgcn_fmed3f(a[1], -57344.f, 57344.f), 0, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(a[2], -57344.f, 57344.f), __builtin_amdgcn_fmed3f(a[3], -57344.f, 57344.f), r, true);
  } else {
    r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(a[1], -448.f, 448.f), 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a[2], -448.f, 448.f), __builtin_amdgcn_fmed3f(a[3], -448.f, 448.f), r, true);
  }
  return (uint32_t)r;
}
// the other direction: channels [8 oc, 8 oc + 8) of voxel v from an e4m3 plane-major tensor [C/16][nvox][16 bytes] (plane bytes per
// 16-channel plane) -- the fp8 mode's activations that were never stored in 16 bits
__device__ __forceinline__ void sp_ld8_e4m3(const void* base, int64_t plane, int64_t v, int oc, float* out) {
  typedef float f2v_ __attribute__((ext_vector_type(2)));
  const uint2 t8 = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(base) + (int64_t)(oc >> 1) * plane + v * 16 + (oc & 1) * 8);
  const f2v_ a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.x, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.x, true);
  const f2v_ a2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.y, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.y, true);
  out[0] = a0[0]; out[1] = a0[1]; out[2] = a1[0]; out[3] = a1[1]; out[4] = a2[0]; out[5] = a2[1]; out[6] = a3[0]; out[7] = a3[1];
}
// channels [8 oc, 8 oc + 8) of voxel v
__device__ __forceinline__ void sp_q8_store8(const SpQ8& q, int64_t v, int oc, const float* vals) {
  *reinterpret_cast<uint2*>(q.p + (int64_t)(oc >> 1) * q.plane + v * 16 + (oc & 1) * 8) =
      make_uint2(sp_q8_pack4(vals, q.scale, q.fmt), sp_q8_pack4(vals + 4, q.scale, q.fmt));
}

// ---- activations (fwd value; derivative as a function of the OUTPUT y) ------------------------
__device__ __forceinline__ float act_fwd(int act, float p, float z) {
  switch (act) {
    case SP_ACT_LEAKY: return z > 0.f ? z : p * z;
    case SP_ACT_ELU: return z > 0.f ? z : p * (__expf(z) - 1.f);
    case SP_ACT_SIGMOID: return 1.f / (1.f + __expf(-z));
    default: return z;
  }
}
// dy/dz from y.  leaky: sign(y)=sign(z) for slope>0.  elu: y<=0 -> y+alpha.  sigmoid: y(1-y)
__device__ __forceinline__ float act_bwd_from_y(int act, float p, float y) {
  switch (act) {
    case SP_ACT_LEAKY: return y > 0.f ? 1.f : p;
    case SP_ACT_ELU: return y > 0.f ? 1.f : y + p;
    case SP_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

// ---- replica rows of the fp64 accumulators the elementwise kernels reduce into (include/stroke_amd.h) ---------------
// sum over the replica rows of accumulator column c (stride <= 0: a single row)
__device__ __forceinline__ double sp_rows_sum(const double* __restrict__ d, int c, int stride) {
  if (stride <= 0) return d[c];
  double t = 0.0;
#pragma unroll
  for (int r = 0; r < SP_REDUCE_ROWS; ++r) t += d[(size_t)r * stride + c];
  return t;
}

// ---- ordered (run-to-run reproducible) block reduction of per-wave column partials: wave w stores its value of column k at
// red[w * ncol + k] (exactly one lane per wave and column), and after a barrier sp_cols_sum adds the waves up in wave order --
// no LDS float atomics, whose arrival order (and with it the fp32 rounding) changes from run to run
__device__ __forceinline__ float sp_cols_sum(const float* red, int ncol, int nwaves, int k) {
  float t = 0.f;
  for (int w = 0; w < nwaves; ++w) t += red[w * ncol + k];
  return t;
}

// ---- exact unsigned division by a runtime constant (host computes mul/shift) --------------------
struct FastDiv { uint32_t mul, shift; };
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv d) { return (__umulhi(n, d.mul) + n) >> d.shift; }
__host__ __device__ static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv r; uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  r.shift = s;
  r.mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
  return r;
}

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the 16 lanes of a DPP row (lanes with equal lane>>4), result in every lane of the row:
// xor-1 / xor-2 inside quads, then the two mirrors -- 4 VALU ops, no LDS crossbar (ds_bpermute) traffic.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// ---- BatchNorm finalize by ONE workgroup (sp_bn_fin_args; blockDim.x a multiple of 32, at most 256; every thread calls it):
// afterwards sc[c] / sh[c] (LDS, CP floats each) hold scale / shift of every channel.  Used by sp_bn_finalize (one workgroup) and
// inside the consumers that fold the result into weights (sp_conv_prep_folded_bn, sp_first_prep_bn: EVERY workgroup derives the
// values -- same loads, same order, same bits -- and the first one publishes them).  Two phases per chunk of 128 channels: 32-lane
// groups split the replica rows (all loads of a thread independent: one memory round trip, whole 512-byte lines per group), then one
// thread per channel adds the groups' partial sums in group order and finalizes in fp64.  (A wave per channel in a loop -- the
// stand-alone kernel's old form, one channel per workgroup -- costs a dependent round trip per channel: 50 us for 96 channels.)
// publish: this workgroup also writes the outputs the backward reads (scale, shift, mean, invstd) and updates the running statistics.
__device__ __forceinline__ void sp_bn_fin_block(const sp_bn_fin_args& f, bool publish, float* sc, float* sh) {
  __shared__ double sp_bn_part_[8][128][2];
  const int t = threadIdx.x, ngrp = blockDim.x >> 5, grp = t >> 5, l32 = t & 31;
  for (int cb = 0; cb < f.CP; cb += 128) {
    const int cn = min(128, f.CP - cb);
    if (f.training) {
      for (int c = l32; c < cn; c += 32) {
        double s1 = 0, s2 = 0;
        if (cb + c < f.C) {      // eight independent 16-byte loads per trip (64 replica rows over 8 groups: one trip)
          const size_t rs = (size_t)f.CP * 2;
          const double* p = f.sums + (size_t)grp * rs + (size_t)(cb + c) * 2;
          int r = grp;
          for (; r + 7 * ngrp < f.nrep; r += 8 * ngrp, p += 8 * ngrp * rs) {
            double2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double2*>(p + (size_t)k * ngrp * rs);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1 += v[k].x; s2 += v[k].y; }
          }
          for (; r < f.nrep; r += ngrp, p += ngrp * rs) { s1 += p[0]; s2 += p[1]; }
        }
        sp_bn_part_[grp][c][0] = s1;
        sp_bn_part_[grp][c][1] = s2;
      }
      __syncthreads();
    }
    for (int c = t; c < cn; c += blockDim.x) {
      const int ch = cb + c;
      float scv = 0.f, shv = 0.f, mean = 0.f, invstd = 0.f;
      if (ch < f.C) {
        if (f.training) {
          double s1 = 0, s2 = 0;
          for (int g = 0; g < ngrp; ++g) { s1 += sp_bn_part_[g][c][0]; s2 += sp_bn_part_[g][c][1]; }
          const double m = s1 / f.count;
          double var = s2 / f.count - m * m;
          if (var < 0) var = 0;
          mean = (float)m;
          invstd = (float)(1.0 / sqrt(var + (double)f.eps));
          if (publish && f.running_mean) {
            const double unb = f.count > 1 ? var * f.count / (f.count - 1) : var;
            f.running_mean[ch] = (1.f - f.momentum) * f.running_mean[ch] + f.momentum * (float)m;
            f.running_var[ch] = (1.f - f.momentum) * f.running_var[ch] + f.momentum * (float)unb;
          }
        } else {
          mean = f.running_mean[ch];
          invstd = 1.f / sqrtf(f.running_var[ch] + f.eps);
        }
        scv = f.gamma[ch] * invstd;
        shv = f.beta[ch] - mean * scv;
      }
      sc[ch] = scv;
      sh[ch] = shv;
      if (publish) {
        f.scale[ch] = scv;
        f.shift[ch] = shv;
        if (f.mean) { f.mean[ch] = mean; f.invstd[ch] = invstd; }
      }
    }
    __syncthreads();
  }
}

// XCD-aware bijective remap of a linear workgroup id: workgroups that share halo data get
// consecutive ids on ONE XCD (b and b+8 share an XCD under round-robin dispatch; speed only).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
  uint32_t q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4): lane l lands at lds_dst + 16*l, lds_dst wave-uniform.
// Issued through inline asm ON PURPOSE: with the builtin the compiler knows that a VMEM op writes LDS and, unable to
// tell the two tile buffers apart, puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows -- inside the MFMA
// loop, where it drains the prefetch of the NEXT tile and serialises DMA and compute (found in the ISA of the
// double-buffered weight-gradient kernel).  The kernels order DMA and reads themselves: s_waitcnt vmcnt(0) + barrier
// before a buffer is read.  m0 is not otherwise used by these kernels (no other LDS-DMA / GWS / movrel).
// Same, WITHOUT the "memory" clobber: for DMAs placed inside an MFMA loop.  The clobber makes the statement a barrier for the
// compiler's own LDS reads and stores, which could then no longer be prefetched / interleaved across it.  Only for DMAs whose
// destination no instruction between the surrounding barriers touches (the kernel's own waits + s_barrier order it).
// -DSP_DMA_NT (diagnostic build, tools/build_variant_all.sh): the non-temporal cache policy on every LDS-DMA load
#ifdef SP_DMA_NT
#define SP_DMA_POLICY " nt"
#else
#define SP_DMA_POLICY ""
#endif
__device__ __forceinline__ void sp_dma16_nc(const void* src, const void* lds_dst) {
  typedef __attribute__((address_space(3))) void sp_lds_void;
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(sp_lds_void*)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" SP_DMA_POLICY ::"v"(src), "s"(base));
}
__device__ __forceinline__ void sp_dma16(const void* src, const void* lds_dst) {
  typedef __attribute__((address_space(3))) void sp_lds_void;
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(sp_lds_void*)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" SP_DMA_POLICY ::"v"(src), "s"(base) : "memory");
}


