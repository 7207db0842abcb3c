// Fused classify head of the U-Net (Unet3D.py:49-54,75-77): Conv3d 1x1x1 (C -> CH) -> LeakyReLU -> Conv3d 1x1x1
// (CH -> NC) -> Sigmoid.  Pure HBM-bound pointwise work (11 and 2 FLOP/B): as generic convolutions it
// took six launches and 574 us per step for 0.8 GFLOP; fused it reads the 16-channel block output once per pass.
//   forward : x (channels-last) -> seg (NCDHW fp32, the layout dto.outputs.core/penu are views of)
//   backward: given dL/dseg, recomputes the hidden layer, writes dz = dL/dx * act'(x) for the producing conv
//             (x is that conv's post-activation output), sum(dz) for its bias gradient, and reduces
//             dW1, db1, dW2, db2 in-kernel (per-wave register tiles over LDS-staged per-voxel vectors,
//             persistent workgroups, one row of partial sums each, added up by sp_head_grad_finish).
#include "sp_common.h"

// weights are read through the constant address space: uniform addresses there always become s_load (the
// parameters are not written while a head kernel runs)
typedef const __attribute__((address_space(4))) float cfloat;

__device__ __forceinline__ bf16x8 hd_tr_read2(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p0));
  bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p1));
  return __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int C, int CH, int NC, typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int64_t nvox_per_b, int64_t total, int CP,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float slope, float* __restrict__ seg, int64_t x_lo = 0) {
  // weights are read at wave-uniform addresses through the constant address space: s_load into SGPR operands
  cfloat *cw1 = (cfloat*)w1, *cb1 = (cfloat*)b1, *cw2 = (cfloat*)w2, *cb2 = (cfloat*)b2;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    float xv[C];
#pragma unroll
    for (int c = 0; c < C; c += 8) ld8x<T>(x + v * CP + c, x_lo, xv + c);
    float o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = cb2[c];
    // rolled on purpose: fully unrolled, all weight s_loads are hoisted to the top and spill (see the backward)
#pragma unroll 1
    for (int k = 0; k < CH; k += 2) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float h = cb1[k + q];
#pragma unroll
        for (int i = 0; i < C; ++i) h = fmaf(cw1[(k + q) * C + i], xv[i], h);
        h = fmaxf(h, slope * h);
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] = fmaf(cw2[c * CH + k + q], h, o[c]);
      }
    }
    const int64_t b = v / nvox_per_b, r = v - b * nvox_per_b;
#pragma unroll
    for (int c = 0; c < NC; ++c) seg[(b * NC + c) * nvox_per_b + r] = 1.f / (1.f + __expf(-o[c]));
  }
}

// LDS record per voxel for the parameter-gradient reduction: dhp[CH], h[CH], x[C], do[NC] (+pad to 4 floats)
template <int C, int CH, int NC, typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ seg,
                                                        const float* __restrict__ dseg, int64_t nvox_per_b, int64_t total,
                                                        int CP, const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, float slope, int act_x, float act_x_p,
                                                        T* __restrict__ dz, float* __restrict__ part) {
  const bool lin_x = act_x == SP_ACT_LEAKY || act_x == SP_ACT_NONE;
  const float slope_x = act_x == SP_ACT_LEAKY ? act_x_p : 1.f;
  constexpr int REC = CH + CH + C + 4;            // floats per voxel record
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sw1 = smem;                               // [CH][C]
  float* sb1 = sw1 + CH * C;                       // [CH]
  float* sw2 = sb1 + CH;                           // [NC][CH]
  float* rec = sw2 + NC * CH;                      // [256][REC]
  for (int i = threadIdx.x; i < CH * C; i += 256) sw1[i] = w1[i];
  for (int i = threadIdx.x; i < CH; i += 256) sb1[i] = b1[i];
  for (int i = threadIdx.x; i < NC * CH; i += 256) sw2[i] = w2[i];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // phase-2 ownership inside a wave: lane -> a 2 x 4 block of dW1 (k0..k0+1, i0..i0+3) when CH*C/8 <= 64 lanes,
  // lane -> one dW2 entry (c2, k2), lanes < CH -> db1[lane], lanes < NC -> db2[lane]
  constexpr int IB = C / 4;                        // i-blocks per k pair
  const int kb = lane / IB, ib = lane - kb * IB;
  const bool own_w1 = kb * 2 < CH;
  const int k0 = own_w1 ? kb * 2 : 0, i0 = ib * 4;
  const bool own_w2 = lane < NC * CH;
  const int c2 = own_w2 ? lane / CH : 0, k2 = own_w2 ? lane - c2 * CH : 0;
  float aw1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float aw2 = 0.f, ab1 = 0.f, ab2 = 0.f;
  float dbz[C];
#pragma unroll
  for (int i = 0; i < C; ++i) dbz[i] = 0.f;
  __syncthreads();

  const int64_t niter = (total + 255) / 256;
  for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
    const int64_t v = it * 256 + tid;
    float* my = rec + tid * REC;
    if (v < total) {
      float xv[C];
#pragma unroll
      for (int c = 0; c < C; c += 8) Store<T>::ld8(x + v * CP + c, xv + c);
      const int64_t b = v / nvox_per_b, r = v - b * nvox_per_b;
      float dov[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int64_t o = (b * NC + c) * nvox_per_b + r;
        const float s = seg[o];
        dov[c] = dseg[o] * s * (1.f - s);
        my[2 * CH + C + c] = dov[c];
      }
      float dx[C];
#pragma unroll
      for (int i = 0; i < C; ++i) { dx[i] = 0.f; my[2 * CH + i] = xv[i]; }
#pragma unroll 4
      for (int k = 0; k < CH; ++k) {
        float hp = sb1[k];
#pragma unroll
        for (int i = 0; i < C; ++i) hp = fmaf(sw1[k * C + i], xv[i], hp);
        float dh = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) dh = fmaf(sw2[c * CH + k], dov[c], dh);
        const float dhp = dh * (hp > 0.f ? 1.f : slope);
        my[k] = dhp;
        my[CH + k] = fmaxf(hp, slope * hp);
#pragma unroll
        for (int i = 0; i < C; ++i) dx[i] = fmaf(sw1[k * C + i], dhp, dx[i]);
      }
#pragma unroll
      for (int i = 0; i < C; ++i) {       // leaky / identity without the per-element switch (wave-uniform `lin_x`)
        dx[i] *= lin_x ? (xv[i] > 0.f ? 1.f : slope_x) : act_bwd_from_y(act_x, act_x_p, xv[i]);
        dbz[i] += dx[i];
      }
#pragma unroll
      for (int c = 0; c < C; c += 8) Store<T>::st8(dz + v * CP + c, dx + c);
      for (int c = C; c < CP; c += 8) { float z8[8] = {0, 0, 0, 0, 0, 0, 0, 0}; Store<T>::st8(dz + v * CP + c, z8); }
    } else {
      for (int i = 0; i < REC; ++i) my[i] = 0.f;
    }
    __syncthreads();
    // phase 2: each wave reduces its own 64 voxel records into the lanes' register tiles
    const float* wr = rec + wave * 64 * REC;
#pragma unroll 4
    for (int u = 0; u < 64; ++u) {
      const float* rv = wr + u * REC;
      if (own_w1) {
        const float d0 = rv[k0], d1 = rv[k0 + 1];
        const float4 xq = *reinterpret_cast<const float4*>(rv + 2 * CH + i0);
        aw1[0][0] = fmaf(d0, xq.x, aw1[0][0]); aw1[0][1] = fmaf(d0, xq.y, aw1[0][1]);
        aw1[0][2] = fmaf(d0, xq.z, aw1[0][2]); aw1[0][3] = fmaf(d0, xq.w, aw1[0][3]);
        aw1[1][0] = fmaf(d1, xq.x, aw1[1][0]); aw1[1][1] = fmaf(d1, xq.y, aw1[1][1]);
        aw1[1][2] = fmaf(d1, xq.z, aw1[1][2]); aw1[1][3] = fmaf(d1, xq.w, aw1[1][3]);
      }
      if (own_w2) aw2 = fmaf(rv[2 * CH + C + c2], rv[CH + k2], aw2);
      if (lane < CH) ab1 += rv[lane];
      if (lane < NC) ab2 += rv[2 * CH + C + lane];
    }
    __syncthreads();
  }
  // flush: one row of partial sums per workgroup, [W1 (CH x C) | b1 (CH) | W2 (NC x CH) | b2 (NC) | sum dz (C)];
  // sp_head_grad_finish adds the rows up (no atomics: thousands of same-address fp64 atomics cost 300 us here)
  constexpr int NQ0 = CH * C + CH + NC * CH + NC, NQ = NQ0 + C;
  float* stage = rec + wave * NQ;                   // the record area is free after the loop's last barrier
  if (own_w1) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) stage[(k0 + a) * C + i0 + j] = aw1[a][j];
  }
  if (lane < CH) stage[CH * C + lane] = ab1;
  if (own_w2) stage[CH * C + CH + c2 * CH + k2] = aw2;
  if (lane < NC) stage[CH * C + CH + NC * CH + lane] = ab2;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const float s = wave_sum(dbz[i]);
    if (lane == 0) stage[NQ0 + i] = s;
  }
  __syncthreads();
  for (int i = tid; i < NQ; i += 256) part[(size_t)blockIdx.x * NQ + i] = rec[i] + rec[NQ + i] + rec[2 * NQ + i] + rec[3 * NQ + i];
}

// ---- bf16 storage: everything on the matrix pipe.  A wave owns 64 voxels per iteration as 4 column groups of 16
// (MFMA column = voxel n = lane % 16, K slice / row quad lg = lane / 16):
//   hp = W1 x      v_mfma 16x16x16, B operand = the lane's own 8 bytes x[v][4lg..4lg+3] straight from global memory;
//   o  = W2 h,  dx = W1^T dhp   chained: the D layout of the first product (rows 4lg..4lg+3 of each 16-row tile)
//                  IS a valid B operand once the weights' K order is permuted to match (no shuffle, no LDS);
//   dW1, db1, dW2  K = voxels: the per-voxel vectors go through per-wave LDS tiles [voxel][16 ch] and come back
//                  transposed (ds_read_b64_tr_b16); AUX = (do_0..do_{NC-1}, 1, 0..) yields dW2 and db1 from one operand.
// Weights enter as hi + lo bf16 pairs (two MFMAs per product): fp32-weight accuracy at bf16 activation precision.
typedef __attribute__((ext_vector_type(4))) short hd_bf16x4;

template <int NP> struct HeadK;      // second-stage operand: 8 (CH = 32) or 4 (CH = 16) K elements per lane
template <> struct HeadK<2> {
  typedef bf16x8 vec;
  static __device__ __forceinline__ int k_of(int lg, int e) { return e < 4 ? 4 * lg + e : 16 + 4 * lg + (e - 4); }
  static __device__ __forceinline__ f32x4 mma(vec a, vec b, f32x4 c) { return SP_MFMA16(a, b, c, 0, 0, 0); }
};
template <> struct HeadK<1> {
  typedef hd_bf16x4 vec;
  static __device__ __forceinline__ int k_of(int lg, int e) { return 4 * lg + e; }
  static __device__ __forceinline__ f32x4 mma(vec a, vec b, f32x4 c) { return SP_MFMA16_K16(a, b, c, 0, 0, 0); }
};
__device__ __forceinline__ void hd_split(float w, short& hi, short& lo) {
  const bf16_t h = f2bf(w);
  hi = (short)h; lo = (short)f2bf(w - bf2f(h));
}

template <int CH, int NC, int CT = 1>      // CT input-channel tiles: C = 16 (the 3-scale network) or 32 (the 4-scale one)
__global__ __launch_bounds__(256) void head_fwd_mfma_kernel(const bf16_t* __restrict__ x, int64_t nvox_per_b, int64_t total,
                                                             int CP, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float slope, float* __restrict__ seg) {
  constexpr int C = 16 * CT, NP = CH / 16, KV = 4 * NP, KX = 4 * CT;
  typedef typename HeadK<NP>::vec kvec;
  typedef typename HeadK<CT>::vec xvec;              // first-stage operands: K = C
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, n = lane & 15;
  xvec a1h[NP], a1l[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int e = 0; e < KX; ++e) { short h, l; hd_split(w1[(16 * t + n) * C + HeadK<CT>::k_of(lg, e)], h, l); a1h[t][e] = h; a1l[t][e] = l; }
  kvec a2h, a2l;
#pragma unroll
  for (int e = 0; e < KV; ++e) { short h, l; hd_split(n < NC ? w2[n * CH + HeadK<NP>::k_of(lg, e)] : 0.f, h, l); a2h[e] = h; a2l[e] = l; }
  float b1k[NP][4];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) b1k[t][j] = b1[16 * t + 4 * lg + j];
  const FastDiv dnv = make_fastdiv((uint32_t)nvox_per_b);      // total < 2^31 (host-checked): 32-bit magic division
  const int64_t niter = (total + 255) / 256;
  for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t v = it * 256 + wave * 64 + g * 16 + n;
      const bool valid = v < total;
      xvec xb;
#pragma unroll
      for (int e = 0; e < KX; ++e) xb[e] = 0;
      if (valid) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const hd_bf16x4 q = *reinterpret_cast<const hd_bf16x4*>(x + v * CP + 16 * ct + 4 * lg);
#pragma unroll
          for (int e = 0; e < 4; ++e) xb[4 * ct + e] = q[e];
        }
      }
      short hb[KV];
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        f32x4 hp = {b1k[t][0], b1k[t][1], b1k[t][2], b1k[t][3]};
        hp = HeadK<CT>::mma(a1h[t], xb, hp);
        hp = HeadK<CT>::mma(a1l[t], xb, hp);
#pragma unroll
        for (int j = 0; j < 4; ++j) hb[4 * t + j] = (short)f2bf(fmaxf(hp[j], slope * hp[j]));
      }
      kvec hv;
#pragma unroll
      for (int e = 0; e < KV; ++e) hv[e] = hb[e];
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = HeadK<NP>::mma(a2h, hv, o);
      o = HeadK<NP>::mma(a2l, hv, o);
      if (valid && lg == 0) {                              // rows 0..3 = classes 0..3 of voxel v
        const int64_t b = fdiv((uint32_t)v, dnv), r = v - b * nvox_per_b;
#pragma unroll
        for (int c = 0; c < NC; ++c) seg[(b * NC + c) * nvox_per_b + r] = 1.f / (1.f + __expf(-(o[c] + b2[c])));
      }
    }
  }
}

template <int CH, int NC, int CT = 1>      // CT input-channel tiles (C = 16 CT)
// (256, 3): without the bound hipcc splits 130 VGPRs + 44 AGPRs (2 waves per SIMD, 116 us); asked for three waves it fits
// everything into 148 VGPRs without scratch: 81 us.  (C = 32: two waves per SIMD.)
__global__ __launch_bounds__(256, CT == 1 ? 3 : 2) void head_bwd_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ seg,
                                                             const float* __restrict__ dseg, int64_t nvox_per_b,
                                                             int64_t total, int CP, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             float slope, int act_x, float act_x_p,
                                                             bf16_t* __restrict__ dz, float* __restrict__ part, const SpQ8 q8) {
  // q8.p != NULL: the fp8 plane-major copy of dz (quantised from the stored 16-bit value, as sp_quantize_f8 would); dz == NULL
  // then skips the 16-bit tensor (both backward convolutions of the producing layer read the copy)
  const bool lin_x = act_x == SP_ACT_LEAKY || act_x == SP_ACT_NONE;
  const float slope_x = act_x == SP_ACT_LEAKY ? act_x_p : 1.f;
  constexpr int C = 16 * CT, NP = CH / 16, NPL = CT + 1 + 2 * NP, KV = 4 * NP, KX = 4 * CT;      // planes: X[CT], AUX, DHP[NP], H[NP]
  constexpr int PLANE = 64 * 32;                                          // bytes of one 64-voxel plane
  typedef typename HeadK<NP>::vec kvec;
  typedef typename HeadK<CT>::vec xvec;
  static_assert(NC <= 3, "AUX row = (do_0..do_{NC-1}, 1) packs into four bf16");
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * NPL * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3, n = li;
  unsigned char* wt = lds + wave * NPL * PLANE;
  unsigned char* tX = wt;
  unsigned char* tA = wt + CT * PLANE;
  unsigned char* tD = wt + (CT + 1) * PLANE;
  unsigned char* tH = wt + (CT + 1 + NP) * PLANE;
  // transposed-read offsets inside a 32-voxel K block (same voxel permutation for both operands)
  const int vq0 = ((lg & 1) ? 2 * lg + 1 : 2 * lg) * 4 + lq;
  const int vq1 = ((lg & 1) ? 2 * lg : 2 * lg + 1) * 4 + lq;
  const int off0 = vq0 * 32 + lp * 8, off1 = vq1 * 32 + lp * 8;

  xvec a1h[NP], a1l[NP];                              // hp = W1 x
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int e = 0; e < KX; ++e) { short h, l; hd_split(w1[(16 * t + n) * C + HeadK<CT>::k_of(lg, e)], h, l); a1h[t][e] = h; a1l[t][e] = l; }
  kvec aTh[CT], aTl[CT];                              // dx = W1^T dhp (one output tile per 16 input channels), K order of the chained operand
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int e = 0; e < KV; ++e) { short h, l; hd_split(w1[HeadK<NP>::k_of(lg, e) * C + 16 * ct + n], h, l); aTh[ct][e] = h; aTl[ct][e] = l; }
  float b1k[NP][4], w2k[NC][NP][4];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b1k[t][j] = b1[16 * t + 4 * lg + j];
#pragma unroll
      for (int c = 0; c < NC; ++c) w2k[c][t][j] = w2[c * CH + 16 * t + 4 * lg + j];
    }

  f32x4 accW1[NP][CT], accW2[NP], accB1[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    accW2[p] = accB1[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) accW1[p][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float dbz[CT][4], db2[NC];                          // dbz: channels 16 ct + 4lg..4lg+3 of this lane's voxels
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) dbz[ct][j] = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) db2[c] = 0.f;

  const FastDiv dnv = make_fastdiv((uint32_t)nvox_per_b);      // total < 2^31 (host-checked): 32-bit magic division
  const int64_t niter = (total + 255) / 256;
  for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t v = it * 256 + wave * 64 + g * 16 + n;
      const bool valid = v < total;
      xvec xb;
#pragma unroll
      for (int e = 0; e < KX; ++e) xb[e] = 0;
      float dov[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) dov[c] = 0.f;
      if (valid) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const hd_bf16x4 q = *reinterpret_cast<const hd_bf16x4*>(x + v * CP + 16 * ct + 4 * lg);
#pragma unroll
          for (int e = 0; e < 4; ++e) xb[4 * ct + e] = q[e];
        }
        const int64_t b = fdiv((uint32_t)v, dnv), r = v - b * nvox_per_b;
        if (NC == 2) {      // the four lanes of a voxel share the loads: even quads read class 0, odd quads class 1
          const int64_t o = (b * NC + (lg & 1)) * nvox_per_b + r;
          const float sg = seg[o];
          dov[lg & 1] = dseg[o] * sg * (1.f - sg);
        } else {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int64_t o = (b * NC + c) * nvox_per_b + r;
            const float sg = seg[o];
            dov[c] = dseg[o] * sg * (1.f - sg);
          }
        }
      }
      if (NC == 2) {        // exchange with the neighbouring quad (lane ^ 16 holds the other class of the same voxel)
        const float mine = (lg & 1) ? dov[NC - 1] : dov[0];
        const float other = __shfl_xor(mine, 16, 64);
        dov[0] = (lg & 1) ? other : mine;
        dov[NC - 1] = (lg & 1) ? mine : other;
      }
      short db_[KV], hb_[KV];
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        f32x4 hp = {b1k[t][0], b1k[t][1], b1k[t][2], b1k[t][3]};
        hp = HeadK<CT>::mma(a1h[t], xb, hp);
        hp = HeadK<CT>::mma(a1l[t], xb, hp);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float dh = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) dh = fmaf(w2k[c][t][j], dov[c], dh);
          db_[4 * t + j] = (short)f2bf(dh * (hp[j] > 0.f ? 1.f : slope));
          hb_[4 * t + j] = (short)f2bf(fmaxf(hp[j], slope * hp[j]));
        }
      }
      kvec dv;
#pragma unroll
      for (int e = 0; e < KV; ++e) dv[e] = db_[e];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 dx = {0.f, 0.f, 0.f, 0.f};
        dx = HeadK<NP>::mma(aTh[ct], dv, dx);
        dx = HeadK<NP>::mma(aTl[ct], dv, dx);
        if (valid) {
          float o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xv = bf2f((bf16_t)xb[4 * ct + j]);
            o4[j] = dx[j] * (lin_x ? (xv > 0.f ? 1.f : slope_x) : act_bwd_from_y(act_x, act_x_p, xv));
            dbz[ct][j] += o4[j];
          }
          if (dz) Store<bf16_t>::st4(dz + v * CP + 16 * ct + 4 * lg, o4);
          if (q8.p) *reinterpret_cast<uint32_t*>(q8.p + (size_t)ct * q8.plane + v * 16 + 4 * lg) = sp_q8_pack4(o4, q8.scale, q8.fmt);
        }
      }
      if (valid && lg == 0) {
        if (dz) for (int c = C; c < CP; c += 8) *reinterpret_cast<uint4*>(dz + v * CP + c) = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) db2[c] += dov[c];
      }
      // ---- this lane's 8 bytes of its voxel's row in every plane
      const int ro = (g * 16 + n) * 32 + 8 * lg;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        hd_bf16x4 x4 = {xb[4 * ct], xb[4 * ct + 1], xb[4 * ct + 2], xb[4 * ct + 3]};
        *reinterpret_cast<hd_bf16x4*>(tX + ct * PLANE + ro) = x4;
      }
      uint32_t a0 = 0, a1 = 0;
      if (lg == 0) {
        float aux[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) aux[c] = dov[c];
        aux[NC] = valid ? 1.f : 0.f;
        a0 = (uint32_t)f2bf(aux[0]) | ((uint32_t)f2bf(aux[1]) << 16);
        a1 = (uint32_t)f2bf(aux[2]) | ((uint32_t)f2bf(aux[3]) << 16);
      }
      *reinterpret_cast<uint2*>(tA + ro) = make_uint2(a0, a1);
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        hd_bf16x4 d4 = {db_[4 * t], db_[4 * t + 1], db_[4 * t + 2], db_[4 * t + 3]};
        hd_bf16x4 h4 = {hb_[4 * t], hb_[4 * t + 1], hb_[4 * t + 2], hb_[4 * t + 3]};
        *reinterpret_cast<hd_bf16x4*>(tD + t * PLANE + ro) = d4;
        *reinterpret_cast<hd_bf16x4*>(tH + t * PLANE + ro) = h4;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int ko = kb * 32 * 32;
      bf16x8 fx[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) fx[ct] = hd_tr_read2(tX + ct * PLANE + ko + off0, tX + ct * PLANE + ko + off1);
      const bf16x8 fa = hd_tr_read2(tA + ko + off0, tA + ko + off1);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const bf16x8 fd = hd_tr_read2(tD + p * PLANE + ko + off0, tD + p * PLANE + ko + off1);
        const bf16x8 fh = hd_tr_read2(tH + p * PLANE + ko + off0, tH + p * PLANE + ko + off1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) accW1[p][ct] = SP_MFMA16(fd, fx[ct], accW1[p][ct], 0, 0, 0);
        accW2[p] = SP_MFMA16(fa, fh, accW2[p], 0, 0, 0);
        accB1[p] = SP_MFMA16(fa, fd, accB1[p], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- flush (MFMA result: lane holds rows lg*4 + j (A side), column li (B side)): one row of partial sums per
  // workgroup, [W1 (CH x C) | b1 (CH) | W2 (NC x CH) | b2 (NC) | sum dz (C)], added up by sp_head_grad_finish
  constexpr int NQ0 = CH * C + CH + NC * CH + NC, NQ = NQ0 + C;
  static_assert(4 * NQ * 4 <= 4 * NPL * PLANE, "per-wave partial rows do not fit the tiles");
  __syncthreads();                                   // every wave has left its tiles
  float* st0 = reinterpret_cast<float*>(lds);
  float* stage = st0 + wave * NQ;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = lg * 4 + j;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) stage[(p * 16 + row) * C + 16 * ct + li] = accW1[p][ct][j];
      if (row < NC) stage[CH * C + CH + row * CH + p * 16 + li] = accW2[p][j];
      if (row == NC) stage[CH * C + p * 16 + li] = accB1[p][j];
    }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float s2 = wave_sum(db2[c]);
    if (lane == 0) stage[CH * C + CH + NC * CH + c] = s2;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sj = row16_sum(dbz[ct][j]);          // over the 16 voxel lanes of this channel quad
      if (n == 0) stage[NQ0 + 16 * ct + 4 * lg + j] = sj;
    }
  __syncthreads();
  for (int i = tid; i < NQ; i += 256) part[(size_t)blockIdx.x * NQ + i] = st0[i] + st0[NQ + i] + st0[2 * NQ + i] + st0[3 * NQ + i];
}

// rows x NQ partial sums -> += into the four parameter gradients and the producing conv's bias-gradient sums
__global__ __launch_bounds__(1024) void head_grad_finish_kernel(const float* __restrict__ part, int rows, int NQ, int nW1, int nb1,
                                                                int nW2, int nb2, float* __restrict__ gW1,
                                                                float* __restrict__ gb1, float* __restrict__ gW2,
                                                                float* __restrict__ gb2, double* __restrict__ dbias_sums) {
  __shared__ double red[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  double acc = 0.0;
  if (col < NQ) {      // (eight independent loads per trip: a plain loop pays a memory round trip per row -- 48 of them)
    int r = rg;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = part[(size_t)(r + 16 * k) * NQ + col];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += (double)t[k];
    }
    for (; r < rows; r += 16) acc += (double)part[(size_t)r * NQ + col];
  }
  red[rg][cl] = acc;
  __syncthreads();
  if (rg == 0 && col < NQ) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    int i = col;
    if (i < nW1) { gW1[i] += (float)t; return; }
    i -= nW1;
    if (i < nb1) { gb1[i] += (float)t; return; }
    i -= nb1;
    if (i < nW2) { gW2[i] += (float)t; return; }
    i -= nW2;
    if (i < nb2) { gb2[i] += (float)t; return; }
    i -= nb2;
    if (dbias_sums) dbias_sums[i] += t;
  }
}

// instantiated shapes: the wave-level reduction tile (f32 kernels) needs CH*C/8 <= 64 and NC*CH <= 64
#define HEAD_CASES(X) X(16, 32, 2) X(16, 16, 2) X(16, 32, 1)
// ... and on the MFMA kernels only (16-bit storage): 32 input channels -- the head of the 4-scale network (32 -> 32 -> 2)
#define HEAD_CASES_MFMA32(X) X(32, 32, 2)

extern "C" int sp_head_supported(int32_t C, int32_t CH, int32_t NC) {
#define X(c, h, n) if (C == c && CH == h && NC == n) return 1;
  HEAD_CASES(X)
#undef X
  return 0;
}
// the same with the storage type: the 32-input-channel shapes exist for SP_BF16 only
extern "C" int sp_head_supported_dtype(int32_t C, int32_t CH, int32_t NC, int32_t dtype) {
  if (sp_head_supported(C, CH, NC)) return 1;
#define X(c, h, n) if (C == c && CH == h && NC == n && dtype == SP_BF16) return 1;
  HEAD_CASES_MFMA32(X)
#undef X
  return 0;
}

extern "C" int sp_head_fwd(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                           const float* b1, int32_t CH, const float* w2, const float* b2, int32_t NC, float slope,
                           float* seg, sp_stream_t stream) {
  SP_CHECK_ARG(x && w1 && b1 && w2 && b2 && seg && CP >= C && CP % 8 == 0, "sp_head_fwd: bad arguments");
  SP_CHECK_ARG((int64_t)B * nvox_per_b < (1ll << 31), "sp_head_fwd: 2^31 voxels or more");
  SP_CHECK_ARG(sp_head_supported_dtype(C, CH, NC, dtype), "sp_head_fwd: no fused kernel for C=%d CH=%d NC=%d dtype=%d", C, CH, NC, dtype);
  const int64_t total = (int64_t)B * nvox_per_b;
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(c, h, n)                                                                                                   \
  if (C == c && CH == h && NC == n) hipLaunchKernelGGL((head_fwd_mfma_kernel<h, n, c / 16>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, nvox_per_b, total, CP, w1, b1, w2, b2, slope, seg);
  HEAD_CASES_MFMA32(X)
#undef X
#define X(c, h, n)                                                                                                   \
  if (C == c && CH == h && NC == n) {                                                                                \
    if (dtype == SP_BF16) hipLaunchKernelGGL((head_fwd_mfma_kernel<h, n>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, nvox_per_b, total, CP, w1, b1, w2, b2, slope, seg); \
    else hipLaunchKernelGGL((head_fwd_kernel<c, h, n, float>), dim3(grid), dim3(256), 0, st, (const float*)x, nvox_per_b, total, CP, w1, b1, w2, b2, slope, seg, (int64_t)0); \
  }
  HEAD_CASES(X)
#undef X
  SP_CHECK_LAUNCH("sp_head_fwd");
  return SP_OK;
}
// bf16 pairs (SP_HL): x = the hi halves of the last block's output, its lo halves x_lo_delta bytes behind; fp32 arithmetic on
// the pair values (the VALU kernel: 16 input channels only)
extern "C" int sp_head_fwd_hl(const void* x, int64_t x_lo_delta, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                              const float* b1, int32_t CH, const float* w2, const float* b2, int32_t NC, float slope,
                              float* seg, sp_stream_t stream) {
  SP_CHECK_ARG(x && x_lo_delta && x_lo_delta % 16 == 0 && w1 && b1 && w2 && b2 && seg && CP >= C && CP % 8 == 0, "sp_head_fwd_hl: bad arguments");
  SP_CHECK_ARG((int64_t)B * nvox_per_b < (1ll << 31), "sp_head_fwd_hl: 2^31 voxels or more");
  SP_CHECK_ARG(sp_head_supported(C, CH, NC), "sp_head_fwd_hl: no fused kernel for C=%d CH=%d NC=%d", C, CH, NC);
  const int64_t total = (int64_t)B * nvox_per_b;
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(c, h, n)                                                                                                   \
  if (C == c && CH == h && NC == n) hipLaunchKernelGGL((head_fwd_kernel<c, h, n, sp_hl_t>), dim3(grid), dim3(256), 0, st, (const sp_hl_t*)x, nvox_per_b, total, CP, w1, b1, w2, b2, slope, seg, x_lo_delta);
  HEAD_CASES(X)
#undef X
  SP_CHECK_LAUNCH("sp_head_fwd_hl");
  return SP_OK;
}

static inline int64_t head_rows(int64_t total) {
  const int64_t niter = (total + 255) / 256;
  return niter < 768 ? niter : 768;
}
extern "C" int64_t sp_head_bwd_rows(int64_t total_voxels) { return head_rows(total_voxels); }
extern "C" int32_t sp_head_row_floats(int32_t C, int32_t CH, int32_t NC) { return CH * C + CH + NC * CH + NC + C; }

static int head_bwd_impl(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                         const float* b1, int32_t CH, const float* w2, int32_t NC, float slope, const float* seg,
                         const float* dseg, int32_t act_x, float act_x_param, void* dz, float* partials, SpQ8 q8,
                         sp_stream_t stream) {
  SP_CHECK_ARG(x && w1 && b1 && w2 && seg && dseg && (dz || q8.p) && partials && CP >= C && CP % 8 == 0, "sp_head_bwd: bad arguments");
  SP_CHECK_ARG(!q8.p || (dtype == SP_BF16 && C % 16 == 0 && q8.scale > 0.f && q8.plane >= (int64_t)B * nvox_per_b * 16), "sp_head_bwd_q8: bf16 storage, whole 16-channel planes");
  SP_CHECK_ARG(sp_head_supported_dtype(C, CH, NC, dtype), "sp_head_bwd: no fused kernel for C=%d CH=%d NC=%d dtype=%d", C, CH, NC, dtype);
  SP_CHECK_ARG(C == 32 || (CH * C / 8 <= 64 && NC * CH <= 64 && CH <= 64), "sp_head_bwd: reduction tile does not fit a wave");
  const int64_t total = (int64_t)B * nvox_per_b;
  SP_CHECK_ARG(total > 0 && total < (1ll << 31), "sp_head_bwd: empty input or 2^31 voxels or more");
  const int rec = CH + CH + C + 4;
  const int lds = (CH * C + CH + NC * CH + 256 * rec) * (int)sizeof(float);
  SP_CHECK_ARG(C == 32 || lds <= 160 * 1024, "sp_head_bwd: LDS %d", lds);
  const unsigned grid = (unsigned)head_rows(total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(c, h, n)                                                                                                   \
  if (C == c && CH == h && NC == n) hipLaunchKernelGGL((head_bwd_mfma_kernel<h, n, c / 16>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, seg, dseg, nvox_per_b, total, CP, w1, b1, w2, slope, act_x, act_x_param, (bf16_t*)dz, partials, q8);
  HEAD_CASES_MFMA32(X)
#undef X
#define X(c, h, n)                                                                                                   \
  if (C == c && CH == h && NC == n) {                                                                                \
    static_assert(c == 16, "head_bwd_mfma_kernel is written for 16 input channels");                                 \
    if (dtype == SP_BF16) {                                                                                          \
      hipLaunchKernelGGL((head_bwd_mfma_kernel<h, n>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, seg, dseg, nvox_per_b, total, CP, w1, b1, w2, slope, act_x, act_x_param, (bf16_t*)dz, partials, q8); \
    } else {                                                                                                         \
      auto kern = head_bwd_kernel<c, h, n, float>;                                                                   \
      SP_ENSURE_LDS(kern, lds, "sp_head_bwd");                                                                       \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const float*)x, seg, dseg, nvox_per_b, total, CP, w1, b1, w2, slope, act_x, act_x_param, (float*)dz, partials); \
    }                                                                                                                \
  }
  HEAD_CASES(X)
#undef X
  SP_CHECK_LAUNCH("sp_head_bwd");
  return SP_OK;
}

extern "C" int sp_head_bwd(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                           const float* b1, int32_t CH, const float* w2, int32_t NC, float slope, const float* seg,
                           const float* dseg, int32_t act_x, float act_x_param, void* dz, float* partials,
                           sp_stream_t stream) {
  return head_bwd_impl(x, dtype, nvox_per_b, B, CP, C, w1, b1, CH, w2, NC, slope, seg, dseg, act_x, act_x_param, dz, partials,
                       SpQ8{nullptr, 0, 1.f, 0}, stream);
}
extern "C" int sp_head_bwd_q8(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                              const float* b1, int32_t CH, const float* w2, int32_t NC, float slope, const float* seg,
                              const float* dseg, int32_t act_x, float act_x_param, void* dz /* or NULL */, float* partials,
                              void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(q8 && (q8_fmt == 0 || q8_fmt == 1), "sp_head_bwd_q8: bad fp8 output");
  return head_bwd_impl(x, dtype, nvox_per_b, B, CP, C, w1, b1, CH, w2, NC, slope, seg, dseg, act_x, act_x_param, dz, partials,
                       SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8_scale, q8_fmt}, stream);
}

extern "C" int sp_head_grad_finish(const float* partials, int64_t rows, int32_t C, int32_t CH, int32_t NC, float* gW1, float* gb1,
                                   float* gW2, float* gb2, double* dbias_sums, sp_stream_t stream) {
  SP_CHECK_ARG(partials && gW1 && gb1 && gW2 && gb2 && rows > 0 && rows <= 768, "sp_head_grad_finish: bad arguments");
  const int NQ = sp_head_row_floats(C, CH, NC);
  hipLaunchKernelGGL(head_grad_finish_kernel, dim3((NQ + 63) / 64), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                     partials, (int)rows, NQ, CH * C, CH, NC * CH, NC, gW1, gb1, gW2, gb2, dbias_sums);
  SP_CHECK_LAUNCH("sp_head_grad_finish");
  return SP_OK;
}
