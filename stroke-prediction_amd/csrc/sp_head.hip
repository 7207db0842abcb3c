// Fused classify head of the U-Net (Unet3D.py:49-54,75-77): Conv3d 1x1x1 (C -> CH) -> LeakyReLU -> Conv3d 1x1x1
// (CH -> NC) -> Sigmoid.  Pure HBM-bound pointwise work (11 and 2 FLOP/B): as generic convolutions it
// took six launches and 574 us per step for 0.8 GFLOP; fused it reads the 16-channel block output once per pass.
//   forward : x (channels-last) -> seg (NCDHW fp32, the layout dto.outputs.core/penu are views of)
//   backward: given dL/dseg, recomputes the hidden layer, writes dz = dL/dx * act'(x) for the producing conv
//             (x is that conv's post-activation output), sum(dz) for its bias gradient, and reduces
//             dW1, db1, dW2, db2 in-kernel (per-wave register tiles over LDS-staged per-voxel vectors,
//             persistent workgroups, one fp64 atomic flush each).
#include "sp_common.h"

template <int C, int CH, int NC, typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int64_t nvox_per_b, int64_t total, int CP,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float slope, float* __restrict__ seg) {
  __shared__ __attribute__((aligned(16))) float sw1[CH * C];
  __shared__ float sb1[CH], sw2[NC * CH], sb2[NC];
  for (int i = threadIdx.x; i < CH * C; i += 256) sw1[i] = w1[i];
  for (int i = threadIdx.x; i < CH; i += 256) sb1[i] = b1[i];
  for (int i = threadIdx.x; i < NC * CH; i += 256) sw2[i] = w2[i];
  for (int i = threadIdx.x; i < NC; i += 256) sb2[i] = b2[i];
  __syncthreads();
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    float xv[C];
#pragma unroll
    for (int c = 0; c < C; c += 8) Store<T>::ld8(x + v * CP + c, xv + c);
    float o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = sb2[c];
#pragma unroll 4
    for (int k = 0; k < CH; ++k) {
      float h = sb1[k];
#pragma unroll
      for (int i = 0; i < C; ++i) h = fmaf(sw1[k * C + i], xv[i], h);
      h = fmaxf(h, slope * h);
#pragma unroll
      for (int c = 0; c < NC; ++c) o[c] = fmaf(sw2[c * CH + k], h, o[c]);
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
                                                        T* __restrict__ dz, double* __restrict__ dbias_sums,
                                                        double* __restrict__ hgrad) {
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
      for (int i = 0; i < C; ++i) { dx[i] *= act_bwd_from_y(act_x, act_x_p, xv[i]); dbz[i] += dx[i]; }
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
  // flush: hgrad layout = [W1 (CH x C) | b1 (CH) | W2 (NC x CH) | b2 (NC)]
  if (own_w1) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&hgrad[(k0 + a) * C + i0 + j], (double)aw1[a][j]);
  }
  if (lane < CH) atomicAdd(&hgrad[CH * C + lane], (double)ab1);
  if (own_w2) atomicAdd(&hgrad[CH * C + CH + c2 * CH + k2], (double)aw2);
  if (lane < NC) atomicAdd(&hgrad[CH * C + CH + NC * CH + lane], (double)ab2);
  if (dbias_sums) {
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const float s = wave_sum(dbz[i]);
      if (lane == 0) atomicAdd(&dbias_sums[i], (double)s);
    }
  }
}

// instantiated shapes: the wave-level reduction tile needs CH*C/8 <= 64 and NC*CH <= 64
#define HEAD_CASES(X) X(16, 32, 2) X(16, 16, 2) X(16, 32, 1)

extern "C" int sp_head_supported(int32_t C, int32_t CH, int32_t NC) {
#define X(c, h, n) if (C == c && CH == h && NC == n) return 1;
  HEAD_CASES(X)
#undef X
  return 0;
}

extern "C" int sp_head_fwd(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                           const float* b1, int32_t CH, const float* w2, const float* b2, int32_t NC, float slope,
                           float* seg, sp_stream_t stream) {
  SP_CHECK_ARG(x && w1 && b1 && w2 && b2 && seg && CP >= C && CP % 8 == 0, "sp_head_fwd: bad arguments");
  SP_CHECK_ARG(sp_head_supported(C, CH, NC), "sp_head_fwd: no fused kernel for C=%d CH=%d NC=%d", C, CH, NC);
  const int64_t total = (int64_t)B * nvox_per_b;
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(c, h, n)                                                                                                   \
  if (C == c && CH == h && NC == n) {                                                                                \
    if (dtype == SP_BF16) hipLaunchKernelGGL((head_fwd_kernel<c, h, n, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, nvox_per_b, total, CP, w1, b1, w2, b2, slope, seg); \
    else hipLaunchKernelGGL((head_fwd_kernel<c, h, n, float>), dim3(grid), dim3(256), 0, st, (const float*)x, nvox_per_b, total, CP, w1, b1, w2, b2, slope, seg); \
  }
  HEAD_CASES(X)
#undef X
  SP_CHECK_LAUNCH("sp_head_fwd");
  return SP_OK;
}

extern "C" int sp_head_bwd(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                           const float* b1, int32_t CH, const float* w2, int32_t NC, float slope, const float* seg,
                           const float* dseg, int32_t act_x, float act_x_param, void* dz, double* dbias_sums,
                           double* hgrad_sums, sp_stream_t stream) {
  SP_CHECK_ARG(x && w1 && b1 && w2 && seg && dseg && dz && hgrad_sums && CP >= C && CP % 8 == 0, "sp_head_bwd: bad arguments");
  SP_CHECK_ARG(sp_head_supported(C, CH, NC), "sp_head_bwd: no fused kernel for C=%d CH=%d NC=%d", C, CH, NC);
  SP_CHECK_ARG(CH * C / 8 <= 64 && NC * CH <= 64 && CH <= 64, "sp_head_bwd: reduction tile does not fit a wave");
  const int64_t total = (int64_t)B * nvox_per_b;
  const int rec = CH + CH + C + 4;
  const int lds = (CH * C + CH + NC * CH + 256 * rec) * (int)sizeof(float);
  SP_CHECK_ARG(lds <= 160 * 1024, "sp_head_bwd: LDS %d", lds);
  const int64_t niter = (total + 255) / 256;
  const unsigned grid = (unsigned)(niter < 256 ? niter : 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(c, h, n)                                                                                                   \
  if (C == c && CH == h && NC == n) {                                                                                \
    if (dtype == SP_BF16) {                                                                                          \
      auto kern = head_bwd_kernel<c, h, n, bf16_t>;                                                                  \
      SP_ENSURE_LDS(kern, lds, "sp_head_bwd");                                                                       \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const bf16_t*)x, seg, dseg, nvox_per_b, total, CP, w1, b1, w2, slope, act_x, act_x_param, (bf16_t*)dz, dbias_sums, hgrad_sums); \
    } else {                                                                                                         \
      auto kern = head_bwd_kernel<c, h, n, float>;                                                                   \
      SP_ENSURE_LDS(kern, lds, "sp_head_bwd");                                                                       \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const float*)x, seg, dseg, nvox_per_b, total, CP, w1, b1, w2, slope, act_x, act_x_param, (float*)dz, dbias_sums, hgrad_sums); \
    }                                                                                                                \
  }
  HEAD_CASES(X)
#undef X
  SP_CHECK_LAUNCH("sp_head_bwd");
  return SP_OK;
}
