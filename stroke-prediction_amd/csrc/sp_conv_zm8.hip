// fp8 (OCP e4m3 / e5m2) output-stationary z-marching convolution: the stride-1 3x3x3 layers of the 4-scale U-Net
// (BASELINE.json configs[4]: "4-scale U-Net ... fp8 MFMA"; topology Unet3D.py:87-146) on v_mfma_f32_16x16x128_f8f6f4.
//
// Replaces nn.Conv3d(3, padding 0) forward and its data gradient (Unet3D.py:19,22) for layers between whole 16-channel planes
// with 32 <= Cin <= 128.  Same march as csrc/sp_conv_zm.hip (read that header first): a workgroup owns a column of NW*MT x 16
// output voxels and walks the INPUT planes; plane i is staged ONCE by LDS-DMA into a ring and feeds the three output planes
// i, i-1, i-2 held in four rotating accumulator sets.  What changes with one-byte operands:
//
//   * activations are stored PLANE-MAJOR in fp8, [Cin/16][B][D][H][W][16 bytes]: a 16-byte DMA chunk is one voxel of one
//     16-channel plane; a ring slot holds half the bytes of the bf16 kernel's;
//   * one MFMA consumes K = 128: each of the four 16-lane groups supplies TWO 16-byte chunks (tap, plane) -- 8 chunks per
//     K step, KS = ceil(9 P / 8) steps per input plane (P = 2: 3 steps, 75 % full; P = 4: 5, 90 %; P = 6: 7, 96 %).  A and B use
//     the same byte order inside a lane group, which is all the instruction asks for (tools/probes/probe_f8.hip);
//   * weights (A operand) are e4m3 with ONE power-of-two scale per output channel (the BatchNorm fold makes their magnitudes
//     channel-dependent; unscaled they would sit in e4m3's subnormal range): sp_conv_prep_f8 packs w * bn_scale[ci] * 2^k[co]
//     and the epilogue multiplies the accumulator by 2^-k[co] (times the data gradient's operand scale, below);
//   * the B operand is e4m3 (forward: activations) or e5m2 (data gradient: dz scaled by a power of two S when it was
//     quantised; 1/S is folded into the epilogue multiplier by the caller);
//   * the epilogue can write, besides the bf16 tensor every elementwise kernel keeps reading, an e4m3 plane-major copy of the
//     output: the next fp8 convolution's operand, produced without another pass over HBM.
//
// K tables (runtime/plan.py:zm8_plan): ktab[(s*4 + g)*2 + h] = byte offset inside a ring slot of chunk h of lane group g in
// step s, ((p*ITH + dy)*18 + dx)*16; weight fragments [(dz*KS + s)*NT + n] of 2 KiB = two 1 KiB halves [h][lane][16 bytes].
#include "sp_common.h"
#include <string.h>

typedef int f8x32 __attribute__((ext_vector_type(8)));     // 32 fp8 values: one lane's share of a K = 128 operand
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvZm8Dev {
  sp_conv_args a;
  const void* zeros;      // >= 16 readable zero bytes
  int32_t nty, ntx;
  uint32_t ncols;
  FastDiv d_tx, d_ty;
};

#define ZM8_SYNC(N)                                                  \
  do {                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");         \
    __builtin_amdgcn_s_barrier();                                    \
    asm volatile("" ::: "memory");                                   \
  } while (0)

__device__ __forceinline__ f8x32 zm8_cat(i32x4 lo, i32x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// four floats -> four e4m3 bytes (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint32_t zm8_pack4_e4m3(const float* v, float s) {
  const float a = __builtin_amdgcn_fmed3f(v[0] * s, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(v[1] * s, -448.f, 448.f);
  const float c = __builtin_amdgcn_fmed3f(v[2] * s, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(v[3] * s, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}

// STATS: per-channel sum / sum of squares of the output (forward layers); ACT: 1 = bias + LeakyReLU / identity, 0 = plain
// (data gradients); BF8: the B operand (x) is e5m2; Q8: also write the e4m3 plane-major copy of the output.
template <int P, int NT, int MT, int NSLOT, int NW, bool STATS, int ACT, bool BF8, bool Q8>
__global__ __launch_bounds__(64 * NW, NW / 4) void conv_zm8_kernel(const ConvZm8Dev Q) {
  constexpr int KS = (9 * P + 7) / 8;             // K steps of 128 (8 chunks) per input plane
  constexpr int ITH = NW * MT + 2, ITW = 18;
  constexpr int PCH = ITH * ITW;                  // 16-byte chunks (voxels) of one 16-channel plane
  constexpr int NCH = P * PCH;
  constexpr int NJ = (NCH + 64 * NW - 1) / (64 * NW);
  constexpr int S = NJ * NW * 1024;               // slot stride in bytes
  constexpr int WOFF = NSLOT * S + NW * 1024;     // weight fragments behind the ring and the dump area
  constexpr int D = NSLOT - 1;
  constexpr int NS = MT * NT * (Q8 ? 2 : 1);      // store instructions of one epilogue
  static_assert(D >= 1 && D <= 3 && (D - 1) * (NJ + NS) <= 63, "counted vmcnt does not fit its 6-bit field");
  constexpr int NWF = 3 * KS * NT;                // weight fragments (2 KiB each)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_conv_args& a = Q.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lv = lane & 15, lg = lane >> 4;
  unsigned char* ring = lds;

  // Output-channel slices in ONE launch (a.nslices > 1): the workgroups of an XCD (blockIdx & 7) split into nslices teams, team
  // member j / nslices of every slice marches over the same range of (column, plane) pairs at the same time -- the slices'
  // 32-channel pieces of an output row meet in that XCD's L2 and leave it as whole lines (one launch per slice writes 64 bytes
  // of every 128 / 192-byte row per pass: measured 30 % slower), and the input planes are fetched from HBM once.
  const int nsl = a.nslices > 1 ? a.nslices : 1;
  uint32_t vb, nvb;
  int sl = 0;
  if (nsl > 1) {
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = (gridDim.x >> 3) / (uint32_t)nsl;      // host: gridDim.x % (8 nsl) == 0
    sl = (int)(j % (uint32_t)nsl);
    vb = xcd * per + j / (uint32_t)nsl;
    nvb = per * 8;
  } else {
    vb = xcd_remap(blockIdx.x, gridDim.x);
    nvb = gridDim.x;
  }
  const int c0s = sl * NT * 16;                    // first output channel of this workgroup's slice
  constexpr int OB = ACT == 2 ? 4 : 2;             // bytes per output element (ACT == 2: fp32 partial sums of an input-channel group)
  unsigned char* const y_sl = reinterpret_cast<unsigned char*>(a.y) + (size_t)c0s * OB;
  unsigned char* const y8_sl = Q8 ? reinterpret_cast<unsigned char*>(a.y8) + (size_t)sl * NT * a.y8_plane : nullptr;
  const float* const bias_sl = a.bias ? a.bias + c0s : nullptr;
  double* const stats_sl = a.stats ? a.stats + (size_t)c0s * 2 : nullptr;

  int kv0[KS], kv1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) { kv0[s] = a.ktab[(s * 4 + lg) * 2]; kv1[s] = a.ktab[(s * 4 + lg) * 2 + 1]; }
  const int vbase0 = (wave * MT * ITW + lv) * 16;
  const unsigned char* wl = lds + WOFF + lane * 16;
  {
    const unsigned char* wf = reinterpret_cast<const unsigned char*>(a.wfrag_hi) + (size_t)sl * a.slice_wfrag_stride;
    for (int f = wave; f < NWF * 2; f += NW) sp_dma16(wf + (size_t)f * 1024 + lane * 16, lds + WOFF + f * 1024);
  }

  // per-lane DMA plan: chunk c = (wave + NW j) * 64 + lane -> (plane p, row vy, voxel vx)
  uint32_t rel[NJ];
  int crd[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + NW * j) * 64 + lane;
    const bool ok = c < NCH;
    const int cc = ok ? c : 0;
    const int p = cc / PCH, vox = cc - p * PCH;
    const int vy = vox / ITW, vx = vox - vy * ITW;
    rel[j] = (uint32_t)p * (uint32_t)a.x_plane + (uint32_t)((vy * a.Wi + vx) * 16);
    crd[j] = vy | (vx << 8) | (ok ? 0 : (1 << 30));
  }
  float bj[NT][4], wi[NT][4], s1[NT][4], s2[NT][4];
  const float* wsc = a.f8_wscale + c0s;           // per-output-channel dequantisation multiplier (2^-k, times 1/S for data gradients)
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bj[n][j] = bias_sl ? bias_sl[n * 16 + lg * 4 + j] : 0.f;
      wi[n][j] = wsc[n * 16 + lg * 4 + j];
      s1[n][j] = s2[n][j] = 0.f;
    }
  const float slope = a.act == SP_ACT_NONE ? 1.f : a.act_param;
  const float q8s = a.y8_scale;
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(Q.zeros);

  const uint64_t T = (uint64_t)Q.ncols * a.Do;
  uint64_t pos = T * vb / nvb;
  const uint64_t pend_pos = T * (vb + 1) / nvb;
  while (pos < pend_pos) {
    const uint32_t col = (uint32_t)(pos / (uint32_t)a.Do);
    const int z0 = (int)(pos - (uint64_t)col * a.Do);
    const int z1 = (int)min((uint64_t)a.Do, (uint64_t)z0 + (pend_pos - pos));
    pos += (uint64_t)(z1 - z0);
    const int nz = z1 - z0, nin = nz + 2;
    uint32_t t = col;
    uint32_t q = fdiv(t, Q.d_tx); const int tx = t - q * Q.ntx; t = q;
    q = fdiv(t, Q.d_ty); const int ty = t - q * Q.nty; const int b = q;
    const int oy0 = ty * (NW * MT), ox0 = tx * 16;
    const int iy0 = oy0 + a.o0H, ix0 = ox0 + a.o0W;
    const unsigned char* xin = reinterpret_cast<const unsigned char*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * 16;
    int vmask = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int vy = crd[j] & 0xff, vx = (crd[j] >> 8) & 0xff;
      if (!(crd[j] >> 30) && (unsigned)(iy0 + vy) < (unsigned)a.Hi && (unsigned)(ix0 + vx) < (unsigned)a.Wi) vmask |= 1 << j;
    }
    const unsigned char* pl_src0 = nullptr;
    unsigned char* pl_dst0 = nullptr;
    int pl_mask = 0;
    bool pl_fill = false;
    auto plane_begin = [&](int i, int slot) {
      const int iz = z0 + a.o0D + i;
      pl_mask = ((unsigned)iz < (unsigned)a.Di) ? vmask : 0;
      pl_src0 = xin + (((int64_t)iz * a.Hi + iy0) * a.Wi + ix0) * 16;
      pl_dst0 = ring + slot * S + wave * 1024;
    };
    auto plane_dma = [&](int j, bool inloop) {
      const unsigned char* src = ((pl_mask >> j) & 1) ? pl_src0 + rel[j] : zsrc;
      unsigned char* dst = pl_dst0 + (pl_fill ? 0 : j * (NW * 1024));
      if (inloop) sp_dma16_nc(src, dst); else sp_dma16(src, dst);
    };
    auto load_plane = [&](int i, int slot) {
      plane_begin(i, slot);
#pragma unroll
      for (int j = 0; j < NJ; ++j) plane_dma(j, false);
    };
    unsigned char* yout = y_sl + (size_t)b * a.YD * a.YH * a.YW * a.CPo * OB;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)yout, 0, a.y ? (int)((uint32_t)a.YD * a.YH * a.YW * a.CPo * (uint32_t)OB) : 0, 0x00020000);      // (y == NULL: no records, every store is dropped -- only the e4m3 copy leaves)
    // the e4m3 copy: plane n of this launch, sample b -- one descriptor per output tile
    __amdgpu_buffer_rsrc_t y8rs[NT];
    if (Q8) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        y8rs[n] = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(y8_sl + (size_t)n * a.y8_plane + (size_t)b * a.YD * a.YH * a.YW * 16), 0,
            (int)((uint32_t)a.YD * a.YH * a.YW * 16u), 0x00020000);
    }
    const int ox = ox0 + lv;
    const bool colok = ox < a.Wo;
    uint32_t rowoff[MT], rowok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int oy = oy0 + wave * MT + m;
      const bool ok = colok && oy < a.Ho;
      rowoff[m] = ok ? (uint32_t)(oy * a.YW + ox) : 0x80000000u;       // voxel index inside an output plane
      rowok[m] = ok ? 0xffffffffu : 0u;
    }
    const uint32_t zvox = (uint32_t)(a.YH * a.YW);

    f32x4 acc[4][NT][MT];

    ZM8_SYNC(0);
#pragma unroll
    for (int k = 0; k < NSLOT - 1; ++k)
      if (k < nin) load_plane(k, k);

#define ZM8_EPILOGUE(R_, fz_)                                                                                     \
  {                                                                                                               \
    const bool pv = (fz_) >= 0;                                                                                   \
    const uint32_t zoff = pv ? (uint32_t)(fz_) * zvox : 0x80000000u;                                              \
    const uint32_t pm = pv ? 0xffffffffu : 0u;                                                                    \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                              \
      const bool inr = !((zoff | rowoff[m]) & 0x80000000u);                                                       \
      const uint32_t vx_ = zoff + rowoff[m];                                                                      \
      const uint32_t off = inr ? vx_ * (uint32_t)(a.CPo * OB) + (uint32_t)(lg * 4 * OB) : 0x80000000u;            \
      const uint32_t off8 = inr ? vx_ * 16u + (uint32_t)(lg * 4) : 0x80000000u;                                   \
      const uint32_t msk = pm & rowok[m];                                                                         \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                            \
        float v[4];                                                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
          if (ACT == 1) { const float zz = fmaf(acc[R_][n][m][j], wi[n][j], bj[n][j]); v[j] = fmaxf(zz, slope * zz); } \
          else v[j] = acc[R_][n][m][j] * wi[n][j];                                                                \
        }                                                                                                         \
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));                                          \
        typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));                                          \
        const u32x2_ d_ = {sp_pack_bf16x2(v[0], v[1]), sp_pack_bf16x2(v[2], v[3])};                               \
        if (ACT == 2) {                                                                                           \
          const u32x4_ f_ = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}; \
          __builtin_amdgcn_raw_buffer_store_b128(f_, yrs, off + (uint32_t)(n * 64), 0, 0);                        \
        } else {                                                                                                  \
          __builtin_amdgcn_raw_buffer_store_b64(d_, yrs, off + (uint32_t)(n * 32), 0, 0);                         \
        }                                                                                                         \
        if (Q8) {      /* of the STORED 16-bit values: the copy equals sp_quantize_f8 of y bit for bit */                \
          const float r_[4] = {sp_h2f_lo(d_.x), sp_h2f_hi(d_.x), sp_h2f_lo(d_.y), sp_h2f_hi(d_.y)};               \
          __builtin_amdgcn_raw_buffer_store_b32(zm8_pack4_e4m3(r_, q8s), y8rs[n], off8, 0, 0);                    \
        }                                                                                                         \
        if (STATS) {                                                                                              \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const float u = __uint_as_float(__float_as_uint(v[j]) & msk);                                         \
            s1[n][j] += u; s2[n][j] = fmaf(u, u, s2[n][j]);                                                       \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }

#define ZM8_LDX(dst, s_)                                                                                          \
  {                                                                                                               \
    const unsigned char* xa_ = sb + vbase0 + kv0[s_];                                                             \
    const unsigned char* xb_ = sb + vbase0 + kv1[s_];                                                             \
    _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                \
        dst[m] = zm8_cat(*reinterpret_cast<const i32x4*>(xa_ + m * (ITW * 16)), *reinterpret_cast<const i32x4*>(xb_ + m * (ITW * 16))); \
  }
#define ZM8_LDW(dst, dz_, s_)                                                                                     \
  _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                                \
    const unsigned char* wp_ = wl + (((dz_) * KS + (s_)) * NT + n) * 2048;                                        \
    dst[n] = zm8_cat(*reinterpret_cast<const i32x4*>(wp_), *reinterpret_cast<const i32x4*>(wp_ + 1024));          \
  }
#define ZM8_DMA(s_) _Pragma("unroll") for (int j = ((s_) * NJ) / KS; j < (((s_) + 1) * NJ) / KS; ++j) plane_dma(j, true);
#define ZM8_MMA(R_, xv_, wv_)                                                                                     \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                  \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                              \
          acc[R_][n][m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv_[n], xv_[m], acc[R_][n][m], 0, BF8 ? 1 : 0, 0, 0, 0, 0);
#define ZM8_MMA0(R_, xv_, wv_)                                                                                    \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                  \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                              \
          acc[R_][n][m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv_[n], xv_[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, BF8 ? 1 : 0, 0, 0, 0, 0);
    // One step = one input plane i (phase PH = i mod 4): taps dz = 0 / 1 / 2 add into the sets PH, PH+3, PH+2 (mod 4) of the
    // output planes i, i-1, i-2; set PH+1 holds plane i-3, whose epilogue is issued among this step's MFMAs.  The K loop runs
    // over (K step s, dz) groups of NT*MT MFMAs; the weight fragments of the NEXT group and the activation fragments of the
    // next K step are read from LDS while the current group computes (two register sets each).
#define ZM8_STEP(PH)                                                                                              \
  {                                                                                                               \
    if (i >= D - 1) ZM8_SYNC((D - 1) * (NJ + NS));                                                                \
    else if (i == 0) ZM8_SYNC((D - 1) * NJ);                                                                      \
    else ZM8_SYNC((D - 1) * NJ + (D > 2 ? 1 : 0) * NS);                                                           \
    pl_fill = false;                                                                                              \
    if (i + D < nin) plane_begin(i + D, (islot + D) % NSLOT);                                                     \
    else { pl_mask = 0; pl_fill = true; pl_dst0 = ring + NSLOT * S + wave * 1024; }                               \
    const unsigned char* sb = ring + islot * S;                                                                   \
    const bool v0 = i < nz, v1 = i >= 1 && i - 1 < nz, v2 = i >= 2 && i - 2 < nz;                                 \
    const int fz = (i >= 3 && i - 3 < nz) ? z0 + i - 3 : -1;                                                      \
    f8x32 xv[2][MT], wv[2][NT];                                                                                   \
    if (v0 && v1 && v2) {                                                                                         \
      ZM8_EPILOGUE((PH + 1) % 4, fz)                                                                              \
      ZM8_LDX(xv[0], 0)                                                                                           \
      ZM8_LDW(wv[0], 2, 0)                                                                                        \
      _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                            \
        _Pragma("unroll") for (int dzi = 0; dzi < 3; ++dzi) {      /* group q = 3 s + dzi: tap dz = 2 - dzi */        \
          const int q_ = 3 * s + dzi;                                                                             \
          if (dzi == 0 && s + 1 < KS) { ZM8_LDX(xv[(s + 1) & 1], s + 1) }                                         \
          if (dzi < 2) { ZM8_LDW(wv[(q_ + 1) & 1], 1 - dzi, s) }                                                  \
          else if (s + 1 < KS) { ZM8_LDW(wv[(q_ + 1) & 1], 2, s + 1) }                                            \
          if (dzi == 0) { ZM8_MMA((PH + 2) % 4, xv[s & 1], wv[q_ & 1]) ZM8_DMA(s) }                               \
          else if (dzi == 1) { ZM8_MMA((PH + 3) % 4, xv[s & 1], wv[q_ & 1]) }                                     \
          else if (s == 0) { ZM8_MMA0(PH, xv[s & 1], wv[q_ & 1]) }                                                \
          else { ZM8_MMA(PH, xv[s & 1], wv[q_ & 1]) }                                                             \
        }                                                                                                         \
      }                                                                                                           \
    } else {                                                                                                      \
      _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                            \
        ZM8_LDX(xv[0], s)                                                                                         \
        ZM8_DMA(s)                                                                                                \
        if (v2) { ZM8_LDW(wv[0], 2, s) ZM8_MMA((PH + 2) % 4, xv[0], wv[0]) }                                      \
        if (v1) { ZM8_LDW(wv[0], 1, s) ZM8_MMA((PH + 3) % 4, xv[0], wv[0]) }                                      \
        if (v0) { ZM8_LDW(wv[0], 0, s) if (s == 0) { ZM8_MMA0(PH, xv[0], wv[0]) } else { ZM8_MMA(PH, xv[0], wv[0]) } } \
      }                                                                                                           \
      ZM8_EPILOGUE((PH + 1) % 4, fz)                                                                              \
    }                                                                                                             \
    ++i;                                                                                                          \
    islot = islot + 1 == NSLOT ? 0 : islot + 1;                                                                   \
  }

    int i = 0, islot = 0;
    while (true) {
      ZM8_STEP(0)
      if (i > nin) break;
      ZM8_STEP(1)
      if (i > nin) break;
      ZM8_STEP(2)
      if (i > nin) break;
      ZM8_STEP(3)
      if (i > nin) break;
    }
#undef ZM8_STEP
#undef ZM8_MMA0
#undef ZM8_MMA
#undef ZM8_DMA
#undef ZM8_LDW
#undef ZM8_LDX
#undef ZM8_EPILOGUE
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (STATS && stats_sl != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);      // [NW waves][NT * 32] (ordered sum: sp_cols_sum)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x1s = row16_sum(s1[n][j]), x2s = row16_sum(s2[n][j]);
        if (lv == 0) { red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2] = x1s; red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2 + 1] = x2s; }
      }
    __syncthreads();
    for (int k = tid; k < NT * 32; k += 64 * NW) {
      const int c = k >> 1;
      if (c0s + c < a.CPo) atomicAdd(&stats_sl[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (k & 1)], (double)sp_cols_sum(red, NT * 32, NW, k));
    }
  }
}

template <int P, int NT, int MT, int NSLOT, int NW, bool STATS, int ACT, bool BF8, bool Q8>
static int launch_zm8_3(const sp_conv_args* a, const void* zeros, hipStream_t st) {
  constexpr int KS = (9 * P + 7) / 8;
  constexpr int NCH = P * (NW * MT + 2) * 18;
  constexpr int NJ = (NCH + 64 * NW - 1) / (64 * NW);
  constexpr int S = NJ * NW * 1024;
  constexpr int lds_bytes = NSLOT * S + NW * 1024 + 3 * KS * NT * 2048;
  static_assert(lds_bytes <= 160 * 1024, "ring + weights do not fit LDS");
  ConvZm8Dev Q;
  Q.a = *a;
  Q.zeros = zeros;
  Q.ntx = (a->Wo + 15) / 16;
  Q.nty = (a->Ho + NW * MT - 1) / (NW * MT);
  Q.ncols = (uint32_t)(a->B * Q.nty * Q.ntx);
  Q.d_tx = make_fastdiv(Q.ntx);
  Q.d_ty = make_fastdiv(Q.nty);
  const uint64_t planes = (uint64_t)Q.ncols * a->Do;
  unsigned grid = planes / 4 < 256 ? (unsigned)(planes / 4 > 0 ? planes / 4 : 1) : 256u;      // one resident workgroup per CU
  if (a->nslices > 1) {      // teams of nslices workgroups per XCD (see the kernel): 8 * nslices * floor(32 / nslices) workgroups
    grid = 8u * (unsigned)a->nslices * (32u / (unsigned)a->nslices);
    SP_CHECK_ARG(planes >= (uint64_t)grid / a->nslices, "sp_conv3d_zm8: too few (column, plane) pairs for %d slices in one launch", a->nslices);
  }
  auto kern = conv_zm8_kernel<P, NT, MT, NSLOT, NW, STATS, ACT, BF8, Q8>;
  SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm8");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
  SP_CHECK_LAUNCH("sp_conv3d_zm8");
  return SP_OK;
}

template <int P, int NT, int MT, int NSLOT, int NW>
static int launch_zm8(const sp_conv_args* a, const void* zeros, hipStream_t st) {
  if (a->dtype_out == SP_F32) {      // fp32 partial sums of one input-channel group (sp_conv_partial_finish adds the groups up)
    if constexpr ((P == 6 && NT == 2) || (P == 8 && NT == 1)) {
      return a->f8_bin ? launch_zm8_3<P, NT, MT, NSLOT, NW, false, 2, true, false>(a, zeros, st)
                       : launch_zm8_3<P, NT, MT, NSLOT, NW, false, 2, false, false>(a, zeros, st);
    } else {
      sp_set_error("sp_conv3d_zm8: the fp32 partial-sum form is built for the (6, 2) and (8, 1) instances");
      return SP_EINVAL;
    }
  }
  if (a->act == SP_ACT_NONE && a->bias == nullptr && a->stats == nullptr && a->y8 == nullptr) {      // data gradients: plain epilogue, bf16 result only
    return a->f8_bin ? launch_zm8_3<P, NT, MT, NSLOT, NW, false, 0, true, false>(a, zeros, st)
                     : launch_zm8_3<P, NT, MT, NSLOT, NW, false, 0, false, false>(a, zeros, st);
  }
  if (a->f8_bin) { sp_set_error("sp_conv3d_zm8: the e5m2 operand form has the plain epilogue only"); return SP_EINVAL; }
  const bool q8 = a->y8 != nullptr;
  if (a->stats) return q8 ? launch_zm8_3<P, NT, MT, NSLOT, NW, true, 1, false, true>(a, zeros, st)
                          : launch_zm8_3<P, NT, MT, NSLOT, NW, true, 1, false, false>(a, zeros, st);
  return q8 ? launch_zm8_3<P, NT, MT, NSLOT, NW, false, 1, false, true>(a, zeros, st)
            : launch_zm8_3<P, NT, MT, NSLOT, NW, false, 1, false, false>(a, zeros, st);
}

// (P input planes of 16 fp8 channels, NT output tiles) -> rows per wave, ring slots, waves per workgroup; SP_EINVAL = no kernel.
// runtime/plan.py (ZM8_CONFIGS) must agree: tests/test_cabi.py checks it.
extern "C" int sp_conv3d_zm8_config(int32_t P, int32_t NT, int32_t* MT, int32_t* NSLOT, int32_t* NW) {
  int mt = 0, ns = 3, nw = 8;
  if (P == 2 && NT == 2) { mt = 2; }
  else if (P == 2 && NT == 1) { mt = 4; }
  else if (P == 4 && NT == 2) { mt = 2; }
  else if (P == 4 && NT == 1) { mt = 4; }
  else if (P == 6 && NT == 2) { mt = 2; ns = 2; }
  else if (P == 8 && NT == 1) { mt = 4; ns = 2; nw = 4; }      // 128 input channels: 9 full K steps, one output tile per launch (weights 54 KiB, two 48 KiB slots)
  if (MT) *MT = mt;
  if (NSLOT) *NSLOT = ns;
  if (NW) *NW = nw;
  return mt ? SP_OK : SP_EINVAL;
}

extern "C" int sp_conv3d_zm8(const sp_conv_args* a, const void* zeros, sp_stream_t stream) {
  SP_CHECK_ARG(a && a->x && (a->y || a->y8) && a->wfrag_hi && a->ktab && a->f8_wscale && zeros && a->in_scale == nullptr, "sp_conv3d_zm8: null pointer (or affine-on-load requested; y may be NULL when the e4m3 copy is asked for)");
  SP_CHECK_ARG((a->dtype_out == SP_BF16 || a->dtype_out == SP_F32) && a->stats_mode == 0 && a->x_plane > 0, "sp_conv3d_zm8: bf16 output (or fp32 partial sums), plain statistics, plane-major fp8 input");
  SP_CHECK_ARG(a->dtype_out == SP_BF16 || (a->bias == nullptr && a->act == SP_ACT_NONE && a->stats == nullptr && a->y8 == nullptr),
               "sp_conv3d_zm8: fp32 partial sums take no bias / activation / statistics / e4m3 copy (sp_conv_partial_finish applies them)");
  SP_CHECK_ARG(a->sD == 1 && a->sH == 1 && a->sW == 1, "sp_conv3d_zm8: stride 1 only");
  SP_CHECK_ARG(a->group_batch == 0, "sp_conv3d_zm8: no BatchNorm groups (run one launch per group)");
  SP_CHECK_ARG(a->act == SP_ACT_LEAKY || a->act == SP_ACT_NONE, "sp_conv3d_zm8: LeakyReLU or identity epilogue");
  SP_CHECK_ARG(a->nslices >= 0 && a->nslices <= 16 && (a->nslices <= 1 || (a->CPo >= a->nslices * a->Cout && a->slice_wfrag_stride > 0 && a->slice_wfrag_stride % 16 == 0)),
               "sp_conv3d_zm8: nslices %d (CPo %d, Cout %d per slice, slice_wfrag_stride %lld)", a->nslices, a->CPo, a->Cout, (long long)a->slice_wfrag_stride);
  SP_CHECK_ARG(a->CPi % 16 == 0 && a->NT == a->NTtot && a->Cout == 16 * a->NT && a->CPo >= a->Cout && a->CPo % 4 == 0,
               "sp_conv3d_zm8: whole 16-channel tiles (CPi %d, Cout %d, NT %d)", a->CPi, a->Cout, a->NT);
  SP_CHECK_ARG(!a->stats || (a->stats_nrep >= 1 && (a->stats_nrep & (a->stats_nrep - 1)) == 0), "sp_conv3d_zm8: stats_nrep must be a power of two");
  SP_CHECK_ARG(a->Do > 0 && a->Ho > 0 && a->Wo > 0 && a->B > 0, "sp_conv3d_zm8: empty output");
  SP_CHECK_ARG(!a->f8_bin || (a->bias == nullptr && a->act == SP_ACT_NONE && a->stats == nullptr && a->y8 == nullptr),
               "sp_conv3d_zm8: the e5m2 (data gradient) form has a plain epilogue");
  const int P = a->CPi / 16;
  SP_CHECK_ARG((uint64_t)P * (uint64_t)a->x_plane < (1ull << 32) && (uint64_t)a->x_plane >= (uint64_t)a->B * a->Di * a->Hi * a->Wi * 16,
               "sp_conv3d_zm8: input planes too large for 32-bit offsets / x_plane smaller than a plane");
  SP_CHECK_ARG((uint64_t)a->YD * a->YH * a->YW * a->CPo * (a->dtype_out == SP_F32 ? 4 : 2) < (1ull << 31), "sp_conv3d_zm8: output sample too large for a buffer descriptor");
  SP_CHECK_ARG(!a->y8 || ((uint64_t)a->y8_plane >= (uint64_t)a->B * a->YD * a->YH * a->YW * 16 && a->y8_scale > 0.f),
               "sp_conv3d_zm8: y8_plane smaller than a plane of the output / y8_scale not positive");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int32_t mt = 0, ns = 0, nw = 0;
  SP_CHECK_ARG(sp_conv3d_zm8_config(P, a->NT, &mt, &ns, &nw) == SP_OK && mt == a->MT, "sp_conv3d_zm8: no kernel for P=%d NT=%d MT=%d", P, a->NT, a->MT);
  if (P == 2 && a->NT == 2) return launch_zm8<2, 2, 2, 3, 8>(a, zeros, st);
  if (P == 2 && a->NT == 1) return launch_zm8<2, 1, 4, 3, 8>(a, zeros, st);
  if (P == 4 && a->NT == 2) return launch_zm8<4, 2, 2, 3, 8>(a, zeros, st);
  if (P == 4 && a->NT == 1) return launch_zm8<4, 1, 4, 3, 8>(a, zeros, st);
  if (P == 6 && a->NT == 2) return launch_zm8<6, 2, 2, 2, 8>(a, zeros, st);
  if (P == 8 && a->NT == 1) return launch_zm8<8, 1, 4, 2, 4>(a, zeros, st);
  return SP_EINVAL;
}

// ---------------------------------------------------------------------------------------------------- input-channel groups
// y[m][c] = act(sum_g partial[g][m][c] + sum_g bias[g][c]) for a convolution whose input channels were split into G groups (more
// input planes than a ring slot holds: the 192 -> 64 and 384 -> 128 layers behind the concatenations, the 256-channel bottleneck
// of Unet3D.py:95-146): one thread = one voxel x 8 channels; statistics of the stored values for the next BatchNorm.
__global__ __launch_bounds__(256) void conv_partial_finish_kernel(const float* __restrict__ partial, int G, int64_t M, int CP,
                                                                   const float* __restrict__ bias, int bias_stride, int act, float ap,
                                                                   bf16_t* __restrict__ y, double* __restrict__ stats, int nrep,
                                                                   unsigned char* __restrict__ y8, int64_t y8_plane) {
  __shared__ __attribute__((aligned(16))) float tr[256 * 16];
  const int OC = CP / 8;
  const int pos = threadIdx.x / OC, oc = threadIdx.x - pos * OC, vpb = 256 / OC;
  const bool active = pos < vpb;
  float bj[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float b = 0.f;
    if (bias && active)
      for (int g = 0; g < G; ++g) b += bias[(size_t)g * bias_stride + oc * 8 + j];
    bj[j] = b; s1[j] = s2[j] = 0.f;
  }
  const float slope = act == SP_ACT_LEAKY ? ap : 1.f;
  if (active) {
    const int64_t chunk = ((M + gridDim.x - 1) / gridDim.x + vpb - 1) / vpb * vpb;
    const int64_t mend = min(M, ((int64_t)blockIdx.x + 1) * chunk);
    // two voxels per trip: 4 G independent 16-byte loads in flight per thread
    for (int64_t m0 = (int64_t)blockIdx.x * chunk + pos; m0 < mend; m0 += 2 * vpb) {
      float v[2][8];
      const bool two = m0 + vpb < mend;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = bj[j];
      const float* p = partial + m0 * CP + oc * 8;
      const size_t ustep = two ? (size_t)vpb * CP : 0;      // (no second voxel: the first one again, not stored)
#pragma unroll 2
      for (int g = 0; g < G; ++g) {
        float4 q[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          q[u][0] = *reinterpret_cast<const float4*>(p + (size_t)g * M * CP + u * ustep);
          q[u][1] = *reinterpret_cast<const float4*>(p + (size_t)g * M * CP + u * ustep + 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          v[u][0] += q[u][0].x; v[u][1] += q[u][0].y; v[u][2] += q[u][0].z; v[u][3] += q[u][0].w;
          v[u][4] += q[u][1].x; v[u][5] += q[u][1].y; v[u][6] += q[u][1].z; v[u][7] += q[u][1].w;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        uint32_t w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z0 = fmaxf(v[u][2 * j], slope * v[u][2 * j]), z1 = fmaxf(v[u][2 * j + 1], slope * v[u][2 * j + 1]);
          w4[j] = sp_pack_bf16x2(z0, z1);
          if (stats) {
            const float q0 = sp_h2f_lo(w4[j]), q1 = sp_h2f_hi(w4[j]);
            s1[2 * j] += q0; s2[2 * j] = fmaf(q0, q0, s2[2 * j]);
            s1[2 * j + 1] += q1; s2[2 * j + 1] = fmaf(q1, q1, s2[2 * j + 1]);
          }
        }
        if (y) *reinterpret_cast<uint4*>(y + (m0 + (int64_t)u * vpb) * CP + oc * 8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);      // (NULL: only the e4m3 copy is wanted)
        if (y8) {      // e4m3 plane-major copy of the stored values: channels [8 oc, 8 oc + 8) = half a 16-byte voxel of plane oc / 2
          const float r0[4] = {sp_h2f_lo(w4[0]), sp_h2f_hi(w4[0]), sp_h2f_lo(w4[1]), sp_h2f_hi(w4[1])};
          const float r1[4] = {sp_h2f_lo(w4[2]), sp_h2f_hi(w4[2]), sp_h2f_lo(w4[3]), sp_h2f_hi(w4[3])};
          *reinterpret_cast<uint2*>(y8 + (size_t)(oc >> 1) * y8_plane + (m0 + (int64_t)u * vpb) * 16 + (oc & 1) * 8) =
              make_uint2(zm8_pack4_e4m3(r0, 1.f), zm8_pack4_e4m3(r1, 1.f));
        }
      }
    }
  }
  if (stats) {      // ordered: thread t's sixteen partials at tr[t][*]; column (channel, moment) = the sum over the voxel slots in order
    float4* d = reinterpret_cast<float4*>(tr + threadIdx.x * 16);
    d[0] = make_float4(s1[0], s2[0], s1[1], s2[1]); d[1] = make_float4(s1[2], s2[2], s1[3], s2[3]);
    d[2] = make_float4(s1[4], s2[4], s1[5], s2[5]); d[3] = make_float4(s1[6], s2[6], s1[7], s2[7]);
    __syncthreads();
    for (int k = threadIdx.x; k < CP * 2; k += 256) {
      float t = 0.f;
      for (int v = 0; v < vpb; ++v) t += tr[(v * OC) * 16 + k];      // thread (v, oc = k / 16) holds columns [16 oc, 16 oc + 16)
      atomicAdd(&stats[(size_t)(blockIdx.x % nrep) * CP * 2 + k], (double)t);
    }
  }
}

extern "C" int sp_conv_partial_finish(const float* partial, int32_t ngroups, int64_t nvox, int32_t CP, const float* bias,
                                      int32_t bias_stride, int32_t act, float act_param, void* y, double* stats, int32_t stats_nrep,
                                      void* y8, int64_t y8_plane, sp_stream_t stream) {
  SP_CHECK_ARG(!y8 || (CP % 16 == 0 && y8_plane >= nvox * 16), "sp_conv_partial_finish: e4m3 copy needs whole 16-channel planes of >= nvox * 16 bytes");
  SP_CHECK_ARG(partial && (y || y8) && ngroups >= 1 && nvox >= 1 && CP % 8 == 0 && CP >= 8 && CP <= 2048, "sp_conv_partial_finish: bad arguments (y may be NULL when the e4m3 copy is asked for)");
  SP_CHECK_ARG(act == SP_ACT_NONE || act == SP_ACT_LEAKY, "sp_conv_partial_finish: LeakyReLU or identity");
  SP_CHECK_ARG(!bias || bias_stride >= CP, "sp_conv_partial_finish: bias_stride");
  SP_CHECK_ARG(!stats || stats_nrep >= 1, "sp_conv_partial_finish: stats replicas");
  const int vpb = 256 / (CP / 8);
  int64_t want = (nvox + (int64_t)vpb * 8 - 1) / ((int64_t)vpb * 8);
  const unsigned grid = (unsigned)(want < 4096 ? (want > 0 ? want : 1) : 4096);
  hipLaunchKernelGGL(conv_partial_finish_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     partial, ngroups, nvox, CP, bias, bias_stride, act, act_param, reinterpret_cast<bf16_t*>(y), stats, stats_nrep,
                     reinterpret_cast<unsigned char*>(y8), y8_plane);
  SP_CHECK_LAUNCH("sp_conv_partial_finish");
  return SP_OK;
}

// ---------------------------------------------------------------------------------------------------- weights
// One workgroup per output channel: amax of the folded row -> power-of-two scale -> e4m3 fragments, the folded bias
// (exact, from the fp32 weights) and the dequantisation multiplier.
__device__ __forceinline__ void prep_f8_row(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin,
                                            const int32_t* __restrict__ kmap, int nsteps, int NT, unsigned char* __restrict__ wfrag,
                                            const float* __restrict__ fold, const float* __restrict__ shift, int ntaps,
                                            const float* __restrict__ bias, float* __restrict__ bias_out, float* __restrict__ winv,
                                            float out_scale, int co) {
  __shared__ float red[8];
  __shared__ float sh_scale;
  const int tid = threadIdx.x;
  float amax = 0.f, bsum = 0.f;
  if (co < Cout)
    for (int i = tid; i < Cin * ntaps; i += 256) {
      const int ci = i / ntaps, tp = i - ci * ntaps;
      const float wv = w[co * sCo + ci * sCi + tp];
      amax = fmaxf(amax, fabsf(wv * (fold ? fold[ci] : 1.f)));
      if (shift) bsum = fmaf(wv, shift[ci], bsum);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, o, 64)); bsum += __shfl_xor(bsum, o, 64); }
  if ((tid & 63) == 0) { red[tid >> 6] = amax; red[4 + (tid >> 6)] = bsum; }
  __syncthreads();
  if (tid == 0) {
    const float am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float bs = red[4] + red[5] + red[6] + red[7];
    // largest power of two that keeps the row inside +-224 (half of e4m3's 448: room for the rounding of the largest element)
    const float sc = (am > 0.f && am < 3.0e38f) ? exp2f(floorf(log2f(224.f / am))) : 1.f;
    sh_scale = sc;
    winv[co] = out_scale / sc;
    if (bias_out) bias_out[co] = co < Cout ? (bias ? bias[co] : 0.f) + bs : 0.f;
  }
  __syncthreads();
  const float sc = sh_scale;
  const int total = nsteps * 8 * 16;            // (step, lane group g, half h, channel c)
  for (int e = tid; e < total; e += 256) {
    const int c = e & 15, ent = e >> 4;          // ent = (step*4 + g)*2 + h
    const int h = ent & 1, g = (ent >> 1) & 3, step = ent >> 3;
    const int km = kmap[ent];
    float v = 0.f;
    if (km >= 0 && co < Cout) {
      const int tap = km >> 16, ci = (km & 0xffff) * 16 + c;
      if (ci < Cin) v = w[co * sCo + ci * sCi + tap] * (fold ? fold[ci] : 1.f) * sc;
    }
    const int r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v, -448.f, 448.f), 0.f, 0, false);
    const size_t f = (size_t)step * NT + (co >> 4);
    wfrag[f * 2048 + h * 1024 + ((g << 4) | (co & 15)) * 16 + c] = (unsigned char)(r & 0xff);
  }
}

__global__ __launch_bounds__(256) void prep_f8_kernel(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin,
                                                       const int32_t* __restrict__ kmap, int nsteps, int NT, unsigned char* __restrict__ wfrag,
                                                       const float* __restrict__ fold, const float* __restrict__ shift, int ntaps,
                                                       const float* __restrict__ bias, float* __restrict__ bias_out, float* __restrict__ winv,
                                                       float out_scale) {
  prep_f8_row(w, sCo, sCi, Cout, Cin, kmap, nsteps, NT, wfrag, fold, shift, ntaps, bias, bias_out, winv, out_scale, blockIdx.x);
}
__global__ __launch_bounds__(256) void prep_f8_batch_kernel(const sp_f8_prep_item* __restrict__ items) {
  const sp_f8_prep_item it = items[blockIdx.y];
  if ((int)blockIdx.x >= it.NT * 16) return;
  prep_f8_row(it.w, it.sCo, it.sCi, it.Cout, it.Cin, it.kmap, it.nsteps, it.NT, reinterpret_cast<unsigned char*>(it.wfrag), it.fold_scale,
              it.fold_shift, it.ntaps, it.bias, it.bias_out, it.winv, it.out_scale, blockIdx.x);
}
extern "C" int sp_conv_prep_f8_batch(const sp_f8_prep_item* items_dev, int32_t n, int32_t max_rows, sp_stream_t stream) {
  SP_CHECK_ARG(items_dev && n >= 1 && n <= 65535 && max_rows >= 16, "sp_conv_prep_f8_batch: bad arguments");
  hipLaunchKernelGGL(prep_f8_batch_kernel, dim3((unsigned)max_rows, (unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), items_dev);
  SP_CHECK_LAUNCH("sp_conv_prep_f8_batch");
  return SP_OK;
}

extern "C" int sp_conv_prep_f8(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap,
                               int32_t nsteps, int32_t NT, void* wfrag, const float* fold_scale, const float* fold_shift,
                               int32_t ntaps, const float* bias, float* bias_out, float* winv, float out_scale,
                               sp_stream_t stream) {
  SP_CHECK_ARG(w && kmap && wfrag && winv && nsteps > 0 && NT > 0 && Cout <= NT * 16 && out_scale > 0.f, "sp_conv_prep_f8: bad arguments");
  hipLaunchKernelGGL(prep_f8_kernel, dim3((unsigned)(NT * 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, sCo, sCi,
                     Cout, Cin, kmap, nsteps, NT, reinterpret_cast<unsigned char*>(wfrag), fold_scale, fold_shift, ntaps, bias,
                     bias_out, winv, out_scale);
  SP_CHECK_LAUNCH("sp_conv_prep_f8");
  return SP_OK;
}

// ---------------------------------------------------------------------------------------------------- activations
// bf16 (channels-last with pitch CP, or plane-major) -> fp8 plane-major [CP/16][nvox][16]; dst = fmt(scale * src)
template <bool BF8>
__global__ __launch_bounds__(256) void quantize_f8_kernel(const bf16_t* __restrict__ src, int CP, int64_t src_plane,
                                                          unsigned char* __restrict__ dst, int64_t dst_plane, int64_t nvox, int P, float scale) {
  const int64_t total = nvox * P;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t vox = idx / P;
    const int p = (int)(idx - vox * P);
    const bf16_t* s = src_plane ? src + (size_t)p * src_plane + (size_t)vox * 16 : src + (size_t)vox * CP + p * 16;
    float v[16];
    Store<bf16_t>::ld8(s, v);
    Store<bf16_t>::ld8(s + 8, v + 8);
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = v[4 * k + j] * scale;
      if (BF8) {
        int r = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(a[0], -57344.f, 57344.f), __builtin_amdgcn_fmed3f(a[1], -57344.f, 57344.f), 0, false);
        r = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(a[2], -57344.f, 57344.f), __builtin_amdgcn_fmed3f(a[3], -57344.f, 57344.f), r, true);
        o[k] = (uint32_t)r;
      } else {
        o[k] = zm8_pack4_e4m3(a, 1.f);
      }
    }
    *reinterpret_cast<uint4*>(dst + (size_t)p * dst_plane + (size_t)vox * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int sp_quantize_f8(const void* src, int32_t CP, int64_t src_plane, void* dst, int64_t dst_plane, int64_t nvox,
                              int32_t fmt, float scale, sp_stream_t stream) {
  SP_CHECK_ARG(src && dst && CP > 0 && CP % 16 == 0 && nvox > 0 && dst_plane >= nvox * 16 && (fmt == 0 || fmt == 1) && scale > 0.f,
               "sp_quantize_f8: bad arguments (CP %d, fmt %d)", CP, fmt);
  const int P = CP / 16;
  const int64_t total = nvox * P;
  const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 32 ? (total + 255) / 256 : 256 * 32);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (fmt) hipLaunchKernelGGL(quantize_f8_kernel<true>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(src), CP, src_plane,
                              reinterpret_cast<unsigned char*>(dst), dst_plane, nvox, P, scale);
  else hipLaunchKernelGGL(quantize_f8_kernel<false>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(src), CP, src_plane,
                          reinterpret_cast<unsigned char*>(dst), dst_plane, nvox, P, scale);
  SP_CHECK_LAUNCH("sp_quantize_f8");
  return SP_OK;
}
