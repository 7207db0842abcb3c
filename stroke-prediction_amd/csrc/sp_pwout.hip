// The decoder's output layer as three streaming kernels (bf16 storage):
//   BatchNorm3d(16) -> Conv3d(16, 1, 1) -> Sigmoid   (Cae3D.py:214-218; also any "n <= 16 channels -> 1 channel" head)
//
// On the generic path this one-output-channel layer costs as much as a 16 -> 16 one, several times over: the normalised input is
// written out (235 MB at 16 x 28 x 128 x 128 voxels), the 1x1x1 kernel writes a 16-channel fp32 tensor for its one channel
// (470 MB) that a transpose then reduces to NCDHW (29 MB), and the backward pads dL/dout to 16 bf16 channels, runs a 16 -> 1
// weight gradient and a 1 -> 16 data gradient over them and reduces the BatchNorm-backward sums in a pass of its own.  Here:
//   forward   out[b, v] = sigmoid(sum_c (w_c s_c) x[b, v, c] + (bias + sum_c w_c t_c))  straight from the RAW input x (the
//             BatchNorm's scale s / shift t of the sample's group folded into 16 coefficients): 32 B in, 4 B out per voxel;
//   backward  dz = dout * out * (1 - out) per voxel; g[b, v, c] = w_c dz (the gradient at the BatchNorm's output, 16 bf16
//             channels: what the previous layer's activation-backward pass consumes) and the 17 sums (sum dz, sum dz * x_c) per
//             group -- everything else is algebra on those: dW_c = s_c sum dz x_c + t_c sum dz, dbias = sum dz, and the
//             BatchNorm-backward pair (sum g_c, sum g_c x_c) = w_c (sum dz, sum dz x_c);
//   finish    that algebra (one workgroup).
#include "sp_common.h"

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define PWO_ROW 32      // doubles per replica row of the backward sums: [0] sum dz, [1 + c] sum dz * x_c

__global__ __launch_bounds__(256) void pwout_fwd_kernel(const bf16_t* __restrict__ x, int64_t V, int Cin, const float* __restrict__ coef,
                                                        int coef_gstride, int group_batch, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out) {
  __shared__ float ws[17];
  const int b = blockIdx.y;
  const float* cf = coef ? coef + (size_t)(group_batch > 0 ? b / group_batch : 0) * coef_gstride : nullptr;      // rows: scale, -, shift (pitch 16)
  if (threadIdx.x < 16) ws[threadIdx.x] = threadIdx.x < Cin ? w[threadIdx.x] * (cf ? cf[threadIdx.x] : 1.f) : 0.f;
  if (threadIdx.x == 16) {
    float s = bias ? bias[0] : 0.f;
    if (cf)
      for (int c = 0; c < Cin; ++c) s = fmaf(w[c], cf[32 + c], s);
    ws[16] = s;
  }
  __syncthreads();
  float wr[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) wr[c] = ws[c];
  const float b0 = ws[16];
  const bf16_t* xb = x + (size_t)b * V * 16;
  float* ob = out + (size_t)b * V;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    float f[16];
    Store<bf16_t>::ld8(xb + v * 16, f);
    Store<bf16_t>::ld8(xb + v * 16 + 8, f + 8);
    float z = b0;
#pragma unroll
    for (int c = 0; c < 16; ++c) z = fmaf(wr[c], f[c], z);
    ob[v] = act_fwd(SP_ACT_SIGMOID, 0.f, z);
  }
}

__global__ __launch_bounds__(256) void pwout_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                        const bf16_t* __restrict__ x, int64_t V, int Cin, const float* __restrict__ w,
                                                        int group_batch, int nrep, bf16_t* __restrict__ g, double* __restrict__ sums) {
  __shared__ float red[4 * 17];      // [wave][sum], added up in wave order (sp_cols_sum)
  const int b = blockIdx.y;
  float wr[16], part[17];
#pragma unroll
  for (int c = 0; c < 16; ++c) wr[c] = c < Cin ? w[c] : 0.f;
#pragma unroll
  for (int k = 0; k < 17; ++k) part[k] = 0.f;
  const bf16_t* xb = x + (size_t)b * V * 16;
  bf16_t* gb = g + (size_t)b * V * 16;
  const float* db = dout + (size_t)b * V;
  const float* ob = out + (size_t)b * V;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const float dz = db[v] * act_bwd_from_y(SP_ACT_SIGMOID, 0.f, ob[v]);
    float f[16], gv[16];
    Store<bf16_t>::ld8(xb + v * 16, f);
    Store<bf16_t>::ld8(xb + v * 16 + 8, f + 8);
    part[0] += dz;
#pragma unroll
    for (int c = 0; c < 16; ++c) { part[1 + c] = fmaf(dz, f[c], part[1 + c]); gv[c] = wr[c] * dz; }
    Store<bf16_t>::st8(gb + v * 16, gv);
    Store<bf16_t>::st8(gb + v * 16 + 8, gv + 8);
  }
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const float s = wave_sum(part[k]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 17 + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 17) {
    const int grp = group_batch > 0 ? b / group_batch : 0;
    atomicAdd(&sums[((size_t)grp * nrep + (blockIdx.x + blockIdx.y) % nrep) * PWO_ROW + threadIdx.x], (double)sp_cols_sum(red, 17, 4, threadIdx.x));
  }
}

// one workgroup of four waves: a wave takes a group at a time, lane r its replica row r (a serial walk over G x nrep rows costs
// 120 us of dependent loads); per group the BatchNorm-backward pair into replica row 0 of bn_sums ([G][bn_nrep][16][2], other
// rows untouched: the caller zeroed them); the weight and bias gradients (+=; NULL: a frozen layer) add the groups up in order
#define PWO_MAXG 16
__global__ __launch_bounds__(256) void pwout_finish_kernel(const double* __restrict__ sums, int nrep, int G, int Cin, const float* __restrict__ w,
                                                           const float* __restrict__ coef, int coef_gstride, double* __restrict__ bn_sums,
                                                           int bn_nrep, float* __restrict__ dw, float* __restrict__ dbias) {
  __shared__ double gdw[PWO_MAXG][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int g = wave; g < G; g += 4) {
    double tot[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) {
      double v = 0.0;
      for (int r = lane; r < nrep; r += 64) v += sums[((size_t)g * nrep + r) * PWO_ROW + k];
      tot[k] = wave_sum_d(v);
    }
    const float* cf = coef ? coef + (size_t)g * coef_gstride : nullptr;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (lane == c) {
        const double wc = c < Cin ? (double)w[c] : 0.0;
        if (bn_sums) {
          bn_sums[((size_t)g * bn_nrep) * 32 + c * 2] = wc * tot[0];
          bn_sums[((size_t)g * bn_nrep) * 32 + c * 2 + 1] = wc * tot[1 + c];
        }
        gdw[g][c] = cf ? (double)cf[c] * tot[1 + c] + (double)cf[32 + c] * tot[0] : tot[1 + c];
      }
    }
    if (lane == 16) gdw[g][16] = tot[0];
  }
  __syncthreads();
  if (threadIdx.x < 17) {
    double t = 0.0;
    for (int g = 0; g < G; ++g) t += gdw[g][threadIdx.x];
    if (threadIdx.x < 16) { if (dw && (int)threadIdx.x < Cin) dw[threadIdx.x] += (float)t; }
    else if (dbias) dbias[0] += (float)t;
  }
}

static unsigned pwo_grid(int64_t V) {
  int64_t gx = (V + 256 * 4 - 1) / (256 * 4);
  return (unsigned)(gx < 1 ? 1 : (gx > 4096 ? 4096 : gx));
}

extern "C" int sp_pwout_fwd(const void* x, int32_t B, int64_t V, int32_t Cin, int32_t CP, const float* coef, int32_t coef_gstride,
                            int32_t group_batch, const float* w, const float* bias, float* out, sp_stream_t stream) {
  SP_CHECK_ARG(x && w && out && B >= 1 && V >= 1 && CP == 16 && Cin >= 1 && Cin <= 16, "sp_pwout_fwd: bf16 input of pitch 16 with 1..16 channels");
  SP_CHECK_ARG(!coef || coef_gstride >= 48, "sp_pwout_fwd: coef rows are (scale, -, shift) of pitch 16 per group");
  SP_CHECK_ARG(group_batch >= 0 && (group_batch == 0 || B % group_batch == 0) && B <= 65535, "sp_pwout_fwd: group_batch %d / batch %d", group_batch, B);
  hipLaunchKernelGGL(pwout_fwd_kernel, dim3(pwo_grid(V), B), dim3(256), 0, ST(stream), (const bf16_t*)x, V, Cin, coef, coef_gstride, group_batch, w, bias, out);
  SP_CHECK_LAUNCH("sp_pwout_fwd");
  return SP_OK;
}

extern "C" int sp_pwout_bwd(const float* dout, const float* out, const void* x, int32_t B, int64_t V, int32_t Cin, int32_t CP, const float* w,
                            int32_t group_batch, int32_t nrep, void* g, double* sums, sp_stream_t stream) {
  SP_CHECK_ARG(dout && out && x && w && g && sums && B >= 1 && V >= 1 && CP == 16 && Cin >= 1 && Cin <= 16 && nrep >= 1,
               "sp_pwout_bwd: bf16 tensors of pitch 16 with 1..16 channels");
  SP_CHECK_ARG(group_batch >= 0 && (group_batch == 0 || B % group_batch == 0) && B <= 65535, "sp_pwout_bwd: group_batch %d / batch %d", group_batch, B);
  hipLaunchKernelGGL(pwout_bwd_kernel, dim3(pwo_grid(V), B), dim3(256), 0, ST(stream), dout, out, (const bf16_t*)x, V, Cin, w, group_batch, nrep, (bf16_t*)g, sums);
  SP_CHECK_LAUNCH("sp_pwout_bwd");
  return SP_OK;
}

extern "C" int sp_pwout_finish(const double* sums, int32_t nrep, int32_t G, int32_t Cin, const float* w, const float* coef, int32_t coef_gstride,
                               double* bn_sums, int32_t bn_nrep, float* dw, float* dbias, sp_stream_t stream) {
  SP_CHECK_ARG(sums && w && nrep >= 1 && G >= 1 && G <= PWO_MAXG && Cin >= 1 && Cin <= 16 && (!bn_sums || bn_nrep >= 1), "sp_pwout_finish: bad arguments (at most %d groups)", PWO_MAXG);
  SP_CHECK_ARG(!coef || coef_gstride >= 48, "sp_pwout_finish: coef rows are (scale, -, shift) of pitch 16 per group");
  hipLaunchKernelGGL(pwout_finish_kernel, dim3(1), dim3(256), 0, ST(stream), sums, nrep, G, Cin, w, coef, coef_gstride, bn_sums, bn_nrep, dw, dbias);
  SP_CHECK_LAUNCH("sp_pwout_finish");
  return SP_OK;
}
