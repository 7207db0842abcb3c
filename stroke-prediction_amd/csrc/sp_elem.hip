// HBM-bound kernels of the U-Net / CAE hot path on gfx950: layout changes, BatchNorm statistics and
// backward pieces, MaxPool3d / trilinear-x2 / crop-skip forward and fused backward, Dice sums, Adam.
// Every kernel moves 16 bytes per lane per access on channels-last tensors (one 8-channel octet of
// one voxel), keeps a thread's octet fixed over its grid-stride loop so per-channel partial sums live
// in registers, and reduces wave -> LDS -> one fp64 atomic per channel per workgroup.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

#include "sp_common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------ core
static thread_local char g_err[512] = "";
void sp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" void sp_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
}
extern "C" int sp_version(void) { return 100; }

#define ST(s) reinterpret_cast<hipStream_t>(s)
// grid cap of the grid-stride elementwise kernels: 4096 measured best (2048: +1 % step time, 8192+: +0.5 %; SP_ELEM_BLOCKS)
static int max_blocks_() { static int v = getenv("SP_ELEM_BLOCKS") ? atoi(getenv("SP_ELEM_BLOCKS")) : 4096; return v; }
#define MAX_BLOCKS max_blocks_()

// thread -> (voxel slot, octet) for CP/8 octets; threads beyond vpb*OC idle
struct OctMap {
  int OC, vpb;
  FastDiv d_oc;
};
static inline OctMap make_octmap(int CP) {
  OctMap m;
  m.OC = CP / 8;
  m.vpb = 256 / m.OC;
  m.d_oc = make_fastdiv(m.OC);
  return m;
}
static inline unsigned grid_for(int64_t nvox, int vpb) {
  int64_t nb = (nvox + vpb - 1) / vpb;
  return (unsigned)(nb < 1 ? 1 : (nb > MAX_BLOCKS ? MAX_BLOCKS : nb));
}

// block-level per-channel reduction of NS partial sums per channel (8 channels per thread; thread = slot * OC + octet).
// Every thread parks its 8*NS partials in LDS (row = slot, conflict-free 16-byte stores), the columns are summed by all
// 256 threads in two conflict-free passes and ONE fp64 atomic per (workgroup, column) goes to memory, into replica row
// (workgroup % SP_REDUCE_ROWS) of out[SP_REDUCE_ROWS][CP*NS]; the consumers add the rows.  Why: (1) LDS float atomics
// (the first version) put 128 threads on each address for 2 octets, ~4096 serialised LDS operations per workgroup;
// (2) fp64 atomics from different workgroups onto one 128-byte line are serialised at ~15-20 ns each, and the 2048
// workgroups of a launch all arrive at the kernel's tail: MaxPool 124^3 x16 took 80 us with statistics against 46 us
// without; 4 replica rows already remove that (47 us).  Replicas interleaved within a line do not help.
template <int NS>
__device__ __forceinline__ void block_channel_reduce(const float part[NS][8], int oc, bool active, int CP,
                                                     double* __restrict__ out, float* red, int pitch = 0) {
  __shared__ __attribute__((aligned(16))) float s_tr[256 * 8 * NS];
  (void)oc; (void)red;
  const int n = CP * NS, OC = CP >> 3;
  const int rows = 256 / OC;                                   // active threads = rows * OC
  if (active) {
    float4* d = reinterpret_cast<float4*>(s_tr + (size_t)threadIdx.x * 8 * NS);
    if (NS == 1) {
      d[0] = make_float4(part[0][0], part[0][1], part[0][2], part[0][3]);
      d[1] = make_float4(part[0][4], part[0][5], part[0][6], part[0][7]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j += 2) d[j >> 1] = make_float4(part[0][j], part[NS - 1][j], part[0][j + 1], part[NS - 1][j + 1]);
    }
  }
  __syncthreads();
  double* o = out + (size_t)(blockIdx.x % SP_REDUCE_ROWS) * (pitch ? pitch : n);      // (pitch: the columns are a slice of wider rows)
  if (n <= 256) {
    const int S = 256 / n, col = threadIdx.x % n, grp = threadIdx.x / n;
    float a0 = 0.f, a1 = 0.f;
    if (grp < S) {
      int k = grp;
      for (; k + S < rows; k += 2 * S) { a0 += s_tr[k * n + col]; a1 += s_tr[(k + S) * n + col]; }
      if (k < rows) a0 += s_tr[k * n + col];
    }
    __syncthreads();
    s_tr[threadIdx.x] = a0 + a1;
    __syncthreads();
    if (threadIdx.x < n) {
      float t = 0.f;
      for (int g = 0; g < S; ++g) t += s_tr[g * n + threadIdx.x];
      atomicAdd(&o[threadIdx.x], (double)t);
    }
  } else {
    for (int col = threadIdx.x; col < n; col += 256) {
      float t = 0.f;
      for (int k = 0; k < rows; ++k) t += s_tr[k * n + col];
      atomicAdd(&o[col], (double)t);
    }
  }
}

// act'(y) with the activation fixed at compile time (ACT >= 0) -- with a run-time `act` hipcc keeps the switch inside the
// innermost loops: one scalar compare-and-branch chain per ELEMENT (pool/skip backward: 405 s_cbranch in the loop body)
template <int ACT> __device__ __forceinline__ float act_bwd_t(int act, float p, float y) {
  return act_bwd_from_y(ACT >= 0 ? ACT : act, p, y);
}
#define SP_ACT_DISPATCH(act, LAUNCH)                 \
  switch (act) {                                      \
    case SP_ACT_LEAKY: { LAUNCH(SP_ACT_LEAKY); } break; \
    case SP_ACT_ELU: { LAUNCH(SP_ACT_ELU); } break;     \
    default: { LAUNCH(-1); } break;                     \
  }

// ------------------------------------------------------------------------------------------------ layout
// one thread = one (voxel, 8-channel octet): wide-channel / few-voxel tensors (the CAE latent, 800 x 100) still
// fill the chip; consecutive threads take consecutive voxels of one octet (coalesced NCDHW reads)
template <typename T>
__global__ void ncdhw_to_cl_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int64_t DHW, int CP,
                                   int64_t total) {
  const int OC = CP / 8;
  const int64_t items = total * OC;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t bo = it / DHW, v = it - bo * DHW;     // bo = b * OC + octet
    const int64_t b = bo / OC;
    const int c0 = (int)(bo - b * OC) * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (c0 + j < C) ? src[(b * C + c0 + j) * DHW + v] : 0.f;
    Store<T>::st8(dst + (b * DHW + v) * CP + c0, f);
  }
}
extern "C" int sp_ncdhw_to_cl(const float* src, void* dst, int32_t dtype, int32_t B, int32_t C, int64_t DHW,
                              int32_t CP, sp_stream_t stream) {
  SP_CHECK_ARG(src && dst && C <= CP && CP % 8 == 0, "sp_ncdhw_to_cl: bad arguments");
  const int64_t total = (int64_t)B * DHW;
  const int64_t items_ = total * (CP / 8);
  const unsigned grid = (unsigned)((items_ + 255) / 256 > 8192 ? 8192 : (items_ + 255) / 256);
  if (dtype == SP_BF16) hipLaunchKernelGGL(ncdhw_to_cl_kernel<bf16_t>, dim3(grid), dim3(256), 0, ST(stream), src, (bf16_t*)dst, C, DHW, CP, total);
  else hipLaunchKernelGGL(ncdhw_to_cl_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), src, (float*)dst, C, DHW, CP, total);
  SP_CHECK_LAUNCH("sp_ncdhw_to_cl");
  return SP_OK;
}

template <typename T>
__global__ void cl_to_ncdhw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int64_t DHW, int CP,
                                   int64_t total) {
  const int OC = (C + 7) / 8;
  const int64_t items = total * OC;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t bo = it / DHW, v = it - bo * DHW;
    const int64_t b = bo / OC;
    const int c0 = (int)(bo - b * OC) * 8;
    float f[8];
    Store<T>::ld8(src + (b * DHW + v) * CP + c0, f);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) dst[(b * C + c0 + j) * DHW + v] = f[j];
  }
}
extern "C" int sp_cl_to_ncdhw(const void* src, float* dst, int32_t dtype, int32_t B, int32_t C, int64_t DHW,
                              int32_t CP, sp_stream_t stream) {
  SP_CHECK_ARG(src && dst && C <= CP && CP % 8 == 0, "sp_cl_to_ncdhw: bad arguments");
  const int64_t total = (int64_t)B * DHW;
  const int64_t items_ = total * ((C + 7) / 8);
  const unsigned grid = (unsigned)((items_ + 255) / 256 > 8192 ? 8192 : (items_ + 255) / 256);
  if (dtype == SP_BF16) hipLaunchKernelGGL(cl_to_ncdhw_kernel<bf16_t>, dim3(grid), dim3(256), 0, ST(stream), (const bf16_t*)src, dst, C, DHW, CP, total);
  else hipLaunchKernelGGL(cl_to_ncdhw_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)src, dst, C, DHW, CP, total);
  SP_CHECK_LAUNCH("sp_cl_to_ncdhw");
  return SP_OK;
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// NS=2: (sum x, sum x^2) of one tensor; two-tensor form: (sum g, sum g*x)
template <typename T, bool TWO>
__global__ __launch_bounds__(256) void bn_sums_kernel(const T* __restrict__ x, const T* __restrict__ g, int64_t nvox,
                                                       int CP, OctMap om, double* __restrict__ sums) {
  extern __shared__ float red[];
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  float part[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = part[1][j] = 0.f;
  if (active) {
    // each workgroup walks ONE contiguous voxel range (neighbouring rows stay in its L1 / the XCD's L2)
    const int64_t chunk_ = ((nvox + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nvox, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      float a[8];
      Store<T>::ld8(x + v * CP + oc * 8, a);
      if (TWO) {
        float b[8];
        Store<T>::ld8(g + v * CP + oc * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) { part[0][j] += b[j]; part[1][j] += b[j] * a[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { part[0][j] += a[j]; part[1][j] += a[j] * a[j]; }
      }
    }
  }
  block_channel_reduce<2>(part, oc, active, CP, sums, red);
}

extern "C" int sp_bn_stats(const void* x, int32_t dtype, int64_t nvox, int32_t CP, double* sums, sp_stream_t stream) {
  SP_CHECK_ARG(x && sums && CP % 8 == 0 && CP >= 8 && CP <= 2048, "sp_bn_stats: bad arguments (CP=%d)", CP);
  OctMap om = make_octmap(CP);
  const unsigned grid = grid_for(nvox, om.vpb * 8);
  const size_t sh = (size_t)CP * 2 * sizeof(float);
  if (dtype == SP_BF16) hipLaunchKernelGGL((bn_sums_kernel<bf16_t, false>), dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)x, (const bf16_t*)nullptr, nvox, CP, om, sums);
  else hipLaunchKernelGGL((bn_sums_kernel<float, false>), dim3(grid), dim3(256), sh, ST(stream), (const float*)x, (const float*)nullptr, nvox, CP, om, sums);
  SP_CHECK_LAUNCH("sp_bn_stats");
  return SP_OK;
}
extern "C" int sp_bn_bwd_reduce(const void* g, const void* x, int32_t dtype, int64_t nvox, int32_t CP, double* sums,
                                sp_stream_t stream) {
  SP_CHECK_ARG(g && x && sums && CP % 8 == 0 && CP >= 8 && CP <= 2048, "sp_bn_bwd_reduce: bad arguments");
  OctMap om = make_octmap(CP);
  const unsigned grid = grid_for(nvox, om.vpb * 8);
  const size_t sh = (size_t)CP * 2 * sizeof(float);
  if (dtype == SP_BF16) hipLaunchKernelGGL((bn_sums_kernel<bf16_t, true>), dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)x, (const bf16_t*)g, nvox, CP, om, sums);
  else hipLaunchKernelGGL((bn_sums_kernel<float, true>), dim3(grid), dim3(256), sh, ST(stream), (const float*)x, (const float*)g, nvox, CP, om, sums);
  SP_CHECK_LAUNCH("sp_bn_bwd_reduce");
  return SP_OK;
}

// nn.BatchNorm3d (training): normalise with biased batch variance, update running stats with the
// unbiased one (momentum); eval: running stats.  Pad channels (c >= C) get scale = shift = 0.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const sp_bn_fin_args f) {
  extern __shared__ float bn_fin_lds[];      // [2][CP]: scale, shift (also what the fused consumers keep in LDS)
  sp_bn_fin_block(f, true, bn_fin_lds, bn_fin_lds + f.CP);
}
extern "C" int sp_bn_finalize(const double* sums, int32_t nrep, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                              int32_t C, int32_t CP, float* scale, float* shift, float* mean, float* invstd,
                              sp_stream_t stream) {
  SP_CHECK_ARG(gamma && beta && scale && shift && C <= CP && CP <= 4096, "sp_bn_finalize: bad arguments");
  SP_CHECK_ARG(training ? (sums != nullptr && count > 0) : (running_mean && running_var), "sp_bn_finalize: missing statistics");
  SP_CHECK_ARG(nrep >= 1, "sp_bn_finalize: nrep");
  sp_bn_fin_args f;
  f.sums = sums; f.gamma = gamma; f.beta = beta; f.running_mean = running_mean; f.running_var = running_var;
  f.scale = scale; f.shift = shift; f.mean = mean; f.invstd = mean ? invstd : nullptr;
  f.count = count; f.momentum = momentum; f.eps = eps; f.nrep = nrep; f.training = training; f.C = C; f.CP = CP;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), (size_t)CP * 2 * sizeof(float), ST(stream), f);
  SP_CHECK_LAUNCH("sp_bn_finalize");
  return SP_OK;
}

// The same for G BatchNorm groups of one launch (batched passes of the CAE: samples [g*Bg, (g+1)*Bg) of the batch are pass g,
// Cae3D.py:105-107,230-233): sums [G][nrep][CP][2]; scale / shift at scale + g*coef_stride (the rows of a [G][3][CP] table),
// mean / invstd [G][CP]; the running statistics take the G momentum updates IN GROUP ORDER (one thread per channel), exactly
// as the reference's G sequential module calls do.
// (round 5: one WAVE per group -- the groups' replica gathers are one memory round trip side by side, not G in a row: 6.6 -> ~4 us
// per launch, 22 launches per CAE step -- then one thread walks the groups in order for the running statistics; same sums, same
// order, same bits as the serial form)
__global__ __launch_bounds__(1024) void bn_finalize_groups_kernel(const double* __restrict__ sums, int nrep, double count, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* running_mean, float* running_var,
                                          float momentum, float eps, int training, int C, int CP, int G, int coef_stride,
                                          float* scale, float* shift, float* mean_out, float* invstd_out) {
  __shared__ double s_m[16], s_var[16];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (c < C && training) {
    for (int g = wv; g < G; g += nw) {
      const double* sg = sums + (size_t)g * nrep * CP * 2;
      double s1 = 0, s2 = 0;
      for (int r = lane; r < nrep; r += 64) { s1 += sg[((size_t)r * CP + c) * 2]; s2 += sg[((size_t)r * CP + c) * 2 + 1]; }
      s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
      if (lane == 0) {
        const double m = s1 / count;
        double var = s2 / count - m * m;
        s_m[g] = m;
        s_var[g] = var < 0 ? 0 : var;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float rm = 0.f, rv = 1.f;
  if (c < C && running_mean) { rm = running_mean[c]; rv = running_var[c]; }
  for (int g = 0; g < G; ++g) {
    float* sc_o = scale + (size_t)g * coef_stride;
    float* sh_o = shift + (size_t)g * coef_stride;
    if (c >= C) {
      sc_o[c] = 0.f; sh_o[c] = 0.f;
      if (mean_out) { mean_out[(size_t)g * CP + c] = 0.f; invstd_out[(size_t)g * CP + c] = 0.f; }
      continue;
    }
    float mean, invstd;
    if (training) {
      const double m = s_m[g], var = s_var[g];
      mean = (float)m;
      invstd = (float)(1.0 / sqrt(var + (double)eps));
      const double unb = count > 1 ? var * count / (count - 1) : var;
      rm = (1.f - momentum) * rm + momentum * (float)m;
      rv = (1.f - momentum) * rv + momentum * (float)unb;
    } else {
      mean = rm;
      invstd = 1.f / sqrtf(rv + eps);
    }
    const float sc = gamma[c] * invstd;
    sc_o[c] = sc;
    sh_o[c] = beta[c] - mean * sc;
    if (mean_out) { mean_out[(size_t)g * CP + c] = mean; invstd_out[(size_t)g * CP + c] = invstd; }
  }
  if (training && running_mean && c < C) { running_mean[c] = rm; running_var[c] = rv; }
}
extern "C" int sp_bn_finalize_groups(const double* sums, int32_t nrep, double count, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                                     int32_t C, int32_t CP, int32_t G, int32_t coef_stride, float* scale, float* shift,
                                     float* mean, float* invstd, sp_stream_t stream) {
  SP_CHECK_ARG(gamma && beta && scale && shift && C <= CP && G >= 1 && coef_stride >= CP, "sp_bn_finalize_groups: bad arguments");
  SP_CHECK_ARG(training ? (sums != nullptr && count > 0) : (running_mean && running_var), "sp_bn_finalize_groups: missing statistics");
  SP_CHECK_ARG(nrep >= 1, "sp_bn_finalize_groups: nrep");
  SP_CHECK_ARG(G <= 16, "sp_bn_finalize_groups: at most 16 groups");
  hipLaunchKernelGGL(bn_finalize_groups_kernel, dim3(CP), dim3(64 * G), 0, ST(stream), sums, nrep, count, gamma, beta,
                     running_mean, running_var, momentum, eps, training, C, CP, G, coef_stride, scale, shift, mean, invstd);
  SP_CHECK_LAUNCH("sp_bn_finalize_groups");
  return SP_OK;
}

// backward: sums [G][nrep][CP][2], mean / invstd [G][CP], coef [G][3][CP]; dgamma / dbeta accumulate the G groups
__global__ __launch_bounds__(1024) void bn_bwd_finalize_groups_kernel(const double* __restrict__ sums, int nrep, double count, const float* __restrict__ gamma,
                                              const float* __restrict__ mean, const float* __restrict__ invstd, int C, int CP, int G,
                                              float* dgamma, float* dbeta, float* coef, float pscale) {
  __shared__ float s_dg[16], s_db[16];      // (one wave per group; thread 0 adds the groups in order: the serial form's sums)
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int g = wv; g < G; g += nw) {
    float* cf = coef + (size_t)g * 3 * CP;
    if (c >= C) { if (lane == 0) { cf[c] = 0.f; cf[CP + c] = 0.f; cf[2 * CP + c] = 0.f; } continue; }
    const double* sg = sums + (size_t)g * nrep * CP * 2;
    double s1 = 0, s2 = 0;
    for (int r = lane; r < nrep; r += 64) { s1 += sg[((size_t)r * CP + c) * 2]; s2 += sg[((size_t)r * CP + c) * 2 + 1]; }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) {
      const double mu = mean[(size_t)g * CP + c], is = invstd[(size_t)g * CP + c], ga = gamma[c];
      const double dg = (s2 - mu * s1) * is, db = s1;
      s_dg[g] = pscale * (float)dg; s_db[g] = pscale * (float)db;
      const double c0 = ga * is, c1 = -ga * is * is * dg / count;
      cf[c] = (float)c0;
      cf[CP + c] = (float)c1;
      cf[2 * CP + c] = (float)(-c0 * db / count - c1 * mu);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && c < C && dgamma) {
    float dg_acc = 0.f, db_acc = 0.f;
    for (int g = 0; g < G; ++g) { dg_acc += s_dg[g]; db_acc += s_db[g]; }
    dgamma[c] += dg_acc; dbeta[c] += db_acc;
  }
}
extern "C" int sp_bn_bwd_finalize_groups(const double* sums, int32_t nrep, double count, const float* gamma, const float* mean,
                                         const float* invstd, int32_t C, int32_t CP, int32_t G, float* dgamma, float* dbeta,
                                         float* coef, float param_grad_scale, sp_stream_t stream) {
  SP_CHECK_ARG(sums && gamma && mean && invstd && coef && count > 0 && G >= 1, "sp_bn_bwd_finalize_groups: bad arguments");
  SP_CHECK_ARG(G <= 16, "sp_bn_bwd_finalize_groups: at most 16 groups");
  hipLaunchKernelGGL(bn_bwd_finalize_groups_kernel, dim3(CP), dim3(64 * G), 0, ST(stream), sums, nrep < 1 ? 1 : nrep, count, gamma, mean,
                     invstd, C, CP, G, dgamma, dbeta, coef, param_grad_scale);
  SP_CHECK_LAUNCH("sp_bn_bwd_finalize_groups");
  return SP_OK;
}

// dgamma = (S2 - mean*S1)*invstd ; dbeta = S1 ; dx = coef0*g + coef1*x + coef2 with
// coef0 = gamma*invstd, coef1 = -gamma*invstd^2*dgamma/N, coef2 = -coef0*dbeta/N - coef1*mean
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, int nrep, double count, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ invstd, int C, int CP,
                                       float* dgamma, float* dbeta, float* coef, float pscale) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= C) { if (lane == 0) { coef[c] = 0.f; coef[CP + c] = 0.f; coef[2 * CP + c] = 0.f; } return; }
  double s1 = 0, s2 = 0;
  for (int r = lane; r < nrep; r += 64) { s1 += sums[((size_t)r * CP + c) * 2]; s2 += sums[((size_t)r * CP + c) * 2 + 1]; }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (lane != 0) return;
  const double mu = mean[c], is = invstd[c], ga = gamma[c];
  const double dg = (s2 - mu * s1) * is, db = s1;
  if (dgamma) { dgamma[c] += pscale * (float)dg; dbeta[c] += pscale * (float)db; }
  const double c0 = ga * is, c1 = -ga * is * is * dg / count;
  coef[c] = (float)c0;
  coef[CP + c] = (float)c1;
  coef[2 * CP + c] = (float)(-c0 * db / count - c1 * mu);
}
extern "C" int sp_bn_bwd_finalize(const double* sums, int32_t nrep, double count, const float* gamma, const float* mean,
                                  const float* invstd, int32_t C, int32_t CP, float* dgamma, float* dbeta, float* coef,
                                  float param_grad_scale, sp_stream_t stream) {
  SP_CHECK_ARG(sums && gamma && mean && invstd && coef && count > 0, "sp_bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(CP), dim3(64), 0, ST(stream), sums, nrep < 1 ? 1 : nrep, count, gamma, mean, invstd, C, CP, dgamma, dbeta, coef, param_grad_scale);
  SP_CHECK_LAUNCH("sp_bn_bwd_finalize");
  return SP_OK;
}

// dz = (coef0*g + coef1*y + coef2) * act'(y)  [coef == NULL: dz = g*act'(y)] ; dbias_sums[c] += sum dz
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                          const float* __restrict__ coef, int64_t nvox, int CP,
                                                          OctMap om, int act, float ap, T* __restrict__ dz,
                                                          double* __restrict__ dbias, const SpQ8 q8, int64_t gvox, int64_t y8_plane = 0) {
  // gvox > 0: voxels [g*gvox, (g+1)*gvox) use the coefficient table coef + g*3*CP (the batched passes of the CAE)
  // y8_plane > 0: y is the e4m3 plane-major copy [CP/16][nvox][16 bytes] of the activations (fp8 mode: no 16-bit tensor was stored)
  extern __shared__ float red[];
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  float part[1][8];
  float c0[8], c1[8], c2[8];
  int64_t gend = gvox > 0 ? 0 : nvox;      // first voxel of the NEXT group: the coefficients in registers hold below it
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    part[0][j] = 0.f;
    c0[j] = (coef && active) ? coef[oc * 8 + j] : 1.f;
    c1[j] = (coef && active) ? coef[CP + oc * 8 + j] : 0.f;
    c2[j] = (coef && active) ? coef[2 * CP + oc * 8 + j] : 0.f;
  }
  if (active) {
    // each workgroup walks ONE contiguous voxel range (neighbouring rows stay in its L1 / the XCD's L2)
    const int64_t chunk_ = ((nvox + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nvox, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      if (v >= gend) {       // (only with gvox > 0) entering another group: its coefficient table
        const int64_t gi = v / gvox;
        gend = (gi + 1) * gvox;
        const float* cg = coef + (size_t)gi * 3 * CP;
#pragma unroll
        for (int j = 0; j < 8; ++j) { c0[j] = cg[oc * 8 + j]; c1[j] = cg[CP + oc * 8 + j]; c2[j] = cg[2 * CP + oc * 8 + j]; }
      }
      float a[8], b[8], o[8];
      Store<T>::ld8(g + v * CP + oc * 8, a);
      if (y8_plane) {
        typedef float f2v_ __attribute__((ext_vector_type(2)));
        const uint2 t8 = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(y) + (int64_t)(oc >> 1) * y8_plane + v * 16 + (oc & 1) * 8);
        const f2v_ a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.x, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.x, true);
        const f2v_ a2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.y, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)t8.y, true);
        b[0] = a0[0]; b[1] = a0[1]; b[2] = a1[0]; b[3] = a1[1]; b[4] = a2[0]; b[5] = a2[1]; b[6] = a3[0]; b[7] = a3[1];
      } else Store<T>::ld8(y + v * CP + oc * 8, b);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = (c0[j] * a[j] + c1[j] * b[j] + c2[j]) * act_bwd_t<ACT>(act, ap, b[j]);
        part[0][j] += o[j];
      }
      if (dz) Store<T>::st8(dz + v * CP + oc * 8, o);      // (NULL: both readers take the fp8 copy)
      if (q8.p) sp_q8_store8(q8, v, oc, o);
    }
  }
  if (dbias) block_channel_reduce<1>(part, oc, active, CP, dbias, red);
}
static int bn_act_bwd_impl(const void* g, const void* y, const float* coef, int32_t dtype, int64_t nvox, int32_t CP,
                           int32_t act, float act_param, void* dz, double* dbias_sums, SpQ8 q8, sp_stream_t stream, int64_t gvox = 0,
                           int64_t y8_plane = 0) {
  SP_CHECK_ARG(g && y && (dz || q8.p) && CP % 8 == 0 && CP <= 2048, "sp_bn_act_bwd: bad arguments");
  SP_CHECK_ARG(y8_plane == 0 || (dtype == SP_BF16 && CP % 16 == 0 && y8_plane >= nvox * 16), "sp_bn_act_bwd_y8: y as e4m3 planes needs bf16 gradients, whole 16-channel planes of >= nvox * 16 bytes");
  SP_CHECK_ARG(gvox == 0 || (coef && gvox > 0 && nvox % gvox == 0), "sp_bn_act_bwd_groups: the groups must tile the tensor");
  SP_CHECK_ARG(!q8.p || (dtype == SP_BF16 && CP % 16 == 0 && q8.plane >= nvox * 16 && q8.scale > 0.f), "sp_bn_act_bwd_q8: bf16 tensors of whole 16-channel planes");
  OctMap om = make_octmap(CP);
  const unsigned grid = grid_for(nvox, om.vpb * 4);
  const size_t sh = (size_t)CP * sizeof(float);
#define SP_L(A_)                                                                                                                   \
  if (dtype == SP_BF16) hipLaunchKernelGGL((bn_act_bwd_kernel<bf16_t, A_>), dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)g, \
                                           (const bf16_t*)y, coef, nvox, CP, om, act, act_param, (bf16_t*)dz, dbias_sums, q8, gvox, y8_plane); \
  else hipLaunchKernelGGL((bn_act_bwd_kernel<float, A_>), dim3(grid), dim3(256), sh, ST(stream), (const float*)g, (const float*)y,   \
                          coef, nvox, CP, om, act, act_param, (float*)dz, dbias_sums, q8, gvox)
  SP_ACT_DISPATCH(act, SP_L)
#undef SP_L
  SP_CHECK_LAUNCH("sp_bn_act_bwd");
  return SP_OK;
}
extern "C" int sp_bn_act_bwd(const void* g, const void* y, const float* coef, int32_t dtype, int64_t nvox, int32_t CP,
                             int32_t act, float act_param, void* dz, double* dbias_sums, sp_stream_t stream) {
  return bn_act_bwd_impl(g, y, coef, dtype, nvox, CP, act, act_param, dz, dbias_sums, SpQ8{nullptr, 0, 1.f, 0}, stream);
}
extern "C" int sp_bn_act_bwd_groups(const void* g, const void* y, const float* coef, int32_t dtype, int64_t nvox, int32_t CP,
                                    int32_t act, float act_param, void* dz, double* dbias_sums, int64_t group_vox, sp_stream_t stream) {
  SP_CHECK_ARG(group_vox > 0, "sp_bn_act_bwd_groups: group_vox");
  return bn_act_bwd_impl(g, y, coef, dtype, nvox, CP, act, act_param, dz, dbias_sums, SpQ8{nullptr, 0, 1.f, 0}, stream, group_vox);
}
extern "C" int sp_bn_act_bwd_q8(const void* g, const void* y, const float* coef, int32_t dtype, int64_t nvox, int32_t CP,
                                int32_t act, float act_param, void* dz, double* dbias_sums, void* q8, int64_t q8_plane,
                                int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(q8 && (q8_fmt == 0 || q8_fmt == 1), "sp_bn_act_bwd_q8: bad fp8 output");
  return bn_act_bwd_impl(g, y, coef, dtype, nvox, CP, act, act_param, dz, dbias_sums,
                         SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8_scale, q8_fmt}, stream);
}
// y given as its e4m3 plane-major copy (the fp8 mode stores no 16-bit output for layers all of whose readers take the copy);
// q8 may be NULL (then dz must not be)
extern "C" int sp_bn_act_bwd_y8(const void* g, const void* y8, int64_t y8_plane, const float* coef, int64_t nvox, int32_t CP,
                                int32_t act, float act_param, void* dz, double* dbias_sums, void* q8, int64_t q8_plane,
                                int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(y8_plane > 0 && (!q8 || q8_fmt == 0 || q8_fmt == 1), "sp_bn_act_bwd_y8: bad arguments");
  return bn_act_bwd_impl(g, y8, coef, SP_BF16, nvox, CP, act, act_param, dz, dbias_sums,
                         SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8 ? q8_scale : 1.f, q8_fmt}, stream, 0, y8_plane);
}

// ------------------------------------------------------------------------------------------------ pool / upsample / crop fwd
struct Dims { int B, D, H, W; };
// flat voxel index -> (b, z, y, x).  Three 64-bit divisions per call cost more ALU than the rest of an elementwise
// kernel's iteration: the magic numbers are built once per thread, a call is then three mul-hi/shift pairs.  Entry
// points reject tensors of 2^31 voxels or more (SP_CHECK_VOX).
struct Unflat {
  int D, H, W;
  FastDiv fD, fH, fW;
  __device__ __forceinline__ Unflat(int D_, int H_, int W_) : D(D_), H(H_), W(W_), fD(make_fastdiv(D_)), fH(make_fastdiv(H_)), fW(make_fastdiv(W_)) {}
  __device__ __forceinline__ void operator()(int64_t v64, int& b, int& z, int& y, int& x) const {
    uint32_t v = (uint32_t)v64;
    uint32_t q = fdiv(v, fW); x = (int)(v - q * W); v = q;
    q = fdiv(v, fH); y = (int)(v - q * H); v = q;
    q = fdiv(v, fD); z = (int)(v - q * D); b = (int)q;
  }
};
#define SP_CHECK_VOX(n, what) SP_CHECK_ARG((int64_t)(n) < (1ll << 31), what ": 2^31 voxels or more")

// The same pass for the batched CAE layers whose weight gradient reads the RAW layer input (BatchNorm folded per group, zero
// padding AFTER the normalisation): with x^ = s x + t inside the volume and 0 in the padding,
//   dW[tap] = s * sum_v dz[v] x[v + tap]  +  t * sum_{v: v + tap inside the input} dz[v],
// so next to dz this variant leaves, per group, the sums of dz over the BORDER CLASSES of the output grid (2 pad + 1 classes per
// axis: o < pad, inside, o >= n - pad; class = (cz * ny + cy) * nx + cx, sp_conv_args.bias_tab) in cls_sums[G][ncls][CP] (fp64,
// zeroed by the caller): any tap's box sum is a sum of classes (sp_wgrad_finish_folded_groups).  Workgroups are group-pure
// (grid.y = group); class sums go through fp64 LDS atomics (sums of fp32 values: exact, hence order-independent).
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_cls_kernel(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ coef,
                                                              int64_t gvox, int CP, OctMap om, int act, float ap, T* __restrict__ dz,
                                                              double* __restrict__ dbias, int D, int H, int W, int pz, int py, int px,
                                                              double* __restrict__ cls_sums) {
  extern __shared__ double s_cls[];      // [ncls][CP]
  const int gi = blockIdx.y;
  const int ny = 2 * py + 1, nx = 2 * px + 1, ncls = (2 * pz + 1) * ny * nx;
  const int mid = (pz * ny + py) * nx + px;
  for (int k = threadIdx.x; k < ncls * CP; k += 256) s_cls[k] = 0.0;
  __syncthreads();
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  float part[1][8];
  float c0[8], c1[8], c2[8];
  const float* cg = coef + (size_t)gi * 3 * CP;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    part[0][j] = 0.f;
    c0[j] = active ? cg[oc * 8 + j] : 1.f;
    c1[j] = active ? cg[CP + oc * 8 + j] : 0.f;
    c2[j] = active ? cg[2 * CP + oc * 8 + j] : 0.f;
  }
  const Unflat unflat(D, H, W);
  int cur = mid;
  auto cls1 = [](int o, int p, int n) { return o < p ? o : (o >= n - p ? p + 1 + o - (n - p) : p); };
  {
    const int64_t chunk_ = ((gvox + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t v0 = (int64_t)gi * gvox, vend_ = v0 + min(gvox, ((int64_t)blockIdx.x + 1) * chunk_);
    int64_t v = v0 + (int64_t)blockIdx.x * chunk_ + slot;
    const int niter = (int)((min(chunk_, gvox - (int64_t)blockIdx.x * chunk_) + om.vpb - 1) / om.vpb);      // the same for every thread
    // (z, y, x) of the walk, advanced by the voxels per workgroup each iteration: no divisions inside the loop
    int bb = 0, z = 0, yy = 0, x = 0;
    if (active && v < vend_) unflat(v, bb, z, yy, x);
    int cx = cls1(x, px, W), czy = cls1(z, pz, D) * ny + cls1(yy, py, H);
    // hand this thread's running sums to the table.  A wave's 64 / OC voxels are consecutive, so it usually changes class as ONE
    // (the walk enters another row class): then the lanes of an octet are added up by shuffles and one lane per octet does the
    // atomics; 128 threads flushing to one address each cost ~1000 serialised LDS operations per event otherwise
    auto flush = [&](bool mine) {
      const bool uni = __all(mine && cur == __builtin_amdgcn_readfirstlane(cur));
      if (uni) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = part[0][j];
          for (int o = 32; o >= om.OC; o >>= 1) t += __shfl_xor(t, o, 64);
          if ((int)(threadIdx.x & 63) < om.OC) atomicAdd(&s_cls[cur * CP + oc * 8 + j], (double)t);
          part[0][j] = 0.f;
        }
      } else if (mine) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { atomicAdd(&s_cls[cur * CP + oc * 8 + j], (double)part[0][j]); part[0][j] = 0.f; }
      }
    };
    const bool pow2 = (om.OC & (om.OC - 1)) == 0 && 64 % om.OC == 0;      // (octets per voxel 1, 2, 4, 8: lanes of an octet = lane % OC)
    for (int it = 0; it < niter; ++it, v += om.vpb) {
      const bool ok = active && v < vend_;
      const int cid = ok ? czy * nx + cx : cur;
      const bool chg = cid != cur;
      if (__any(chg)) {
        if (pow2) flush(chg);
        else if (chg) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { atomicAdd(&s_cls[cur * CP + oc * 8 + j], (double)part[0][j]); part[0][j] = 0.f; }
        }
        cur = cid;
      }
      if (ok) {
        float a[8], b[8], o[8];
        Store<T>::ld8(g + v * CP + oc * 8, a);
        Store<T>::ld8(y + v * CP + oc * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (c0[j] * a[j] + c1[j] * b[j] + c2[j]) * act_bwd_t<ACT>(act, ap, b[j]);
        // the STORED values (what the weight gradient multiplies) are what is summed: pack once, store, unpack
        uint32_t wq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[j] = sp_pack_bf16x2(o[2 * j], o[2 * j + 1]);
        *reinterpret_cast<uint4*>(dz + v * CP + oc * 8) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[0][2 * j] += sp_h2f_lo(wq[j]); part[0][2 * j + 1] += sp_h2f_hi(wq[j]); }
      }
      // next voxel of the walk
      x += om.vpb;
      if (x >= W) {
        int dy = 0;
        while (x >= W) { x -= W; ++dy; }
        cx = cls1(x, px, W);
        yy += dy;
        while (yy >= H) { yy -= H; ++z; }
        while (z >= D) z -= D;       // (the next sample(s) of the group: the classes depend on (z, y, x) only; tiny volumes may wrap more than once)
        czy = cls1(z, pz, D) * ny + cls1(yy, py, H);
      }
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&s_cls[cur * CP + oc * 8 + j], (double)part[0][j]);
    }
  }
  __syncthreads();
  double* out = cls_sums + (size_t)gi * ncls * CP;
  for (int k = threadIdx.x; k < ncls * CP; k += 256) {
    const double v = s_cls[k];
    if (v != 0.0) atomicAdd(&out[k], v);
  }
  // the bias gradient's sums (all groups, replica rows as bn_act_bwd): the column totals of this workgroup's table
  if (dbias) {
    for (int c = threadIdx.x; c < CP; c += 256) {
      double t = 0.0;
      for (int cl = 0; cl < ncls; ++cl) t += s_cls[cl * CP + c];
      atomicAdd(&dbias[(size_t)((blockIdx.x + blockIdx.y) % SP_REDUCE_ROWS) * CP + c], t);
    }
  }
}
extern "C" int sp_bn_act_bwd_groups_cls(const void* g, const void* y, const float* coef, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W,
                                        int32_t CP, int32_t act, float act_param, void* dz, double* dbias_sums, int32_t group_batch,
                                        int32_t padD, int32_t padH, int32_t padW, double* cls_sums, sp_stream_t stream) {
  SP_CHECK_ARG(g && y && coef && dz && cls_sums && dtype == SP_BF16 && CP % 8 == 0 && CP <= 64, "sp_bn_act_bwd_groups_cls: bf16 tensors of at most 64 channels");
  SP_CHECK_ARG(group_batch > 0 && B % group_batch == 0 && D >= 1 && H >= 1 && W >= 1, "sp_bn_act_bwd_groups_cls: the groups must tile the batch");
  SP_CHECK_ARG(padD >= 0 && padD <= 2 && padH >= 0 && padH <= 2 && padW >= 0 && padW <= 2 && D >= 2 * padD && H >= 2 * padH && W >= 2 * padW &&
               (2 * padD + 1) * (2 * padH + 1) * (2 * padW + 1) <= 75, "sp_bn_act_bwd_groups_cls: padding 0..2, at most 75 classes");
  const int64_t gvox = (int64_t)group_batch * D * H * W;
  SP_CHECK_VOX((int64_t)B * D * H * W, "sp_bn_act_bwd_groups_cls");
  const int G = B / group_batch, ncls = (2 * padD + 1) * (2 * padH + 1) * (2 * padW + 1);
  OctMap om = make_octmap(CP);
  // few, long workgroups: the class table (zeroing, flush, column totals: ~3 ncls CP LDS operations) is a fixed cost per workgroup
  unsigned gx = grid_for(gvox, om.vpb * 4);
  const unsigned cap = (unsigned)(2048 / G > 0 ? 2048 / G : 1);
  if (gx > cap) gx = cap;
  const size_t sh = (size_t)ncls * CP * sizeof(double);
#define SP_L(A_)                                                                                                                       \
  hipLaunchKernelGGL((bn_act_bwd_cls_kernel<bf16_t, A_>), dim3(gx, G), dim3(256), sh, ST(stream), (const bf16_t*)g, (const bf16_t*)y, coef, gvox, \
                     CP, om, act, act_param, (bf16_t*)dz, dbias_sums, D, H, W, padD, padH, padW, cls_sums)
  SP_ACT_DISPATCH(act, SP_L)
#undef SP_L
  SP_CHECK_LAUNCH("sp_bn_act_bwd_groups_cls");
  return SP_OK;
}

// MaxPool3d(2,2), floor mode (Unet3D.py:39,41)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, Dims di, int CP,
                                                            OctMap om, double* __restrict__ stats, const SpQ8 q8,
                                                            int64_t x_lo = 0, int64_t y_lo = 0, int64_t x8_plane = 0) {
  // x8_plane > 0: x is the e4m3 plane-major copy of the input (the fp8 mode stored no 16-bit tensor); y may then be NULL
  extern __shared__ float red[];
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  const int Do = di.D / 2, Ho = di.H / 2, Wo = di.W / 2;
  const int64_t nout = (int64_t)di.B * Do * Ho * Wo;
  float part[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = part[1][j] = 0.f;
  const Unflat uf_(Do, Ho, Wo);
  if (active) {
    // each workgroup walks ONE contiguous voxel range (neighbouring rows stay in its L1 / the XCD's L2)
    const int64_t chunk_ = ((nout + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nout, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      int b, z, yy, xx;
      uf_(v, b, z, yy, xx);
      float m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = 2 * z + (k >> 2), iy = 2 * yy + ((k >> 1) & 1), ix = 2 * xx + (k & 1);
        float a[8];
        const int64_t vi_ = (((int64_t)b * di.D + iz) * di.H + iy) * di.W + ix;
        if (x8_plane) sp_ld8_e4m3(x, x8_plane, vi_, oc, a);
        else ld8x<T>(x + vi_ * CP + oc * 8, x_lo, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], a[j]);
      }
      if (y) st8x<T>(y + v * CP + oc * 8, y_lo, m);
      if (q8.p) sp_q8_store8(q8, v, oc, m);
#pragma unroll
      for (int j = 0; j < 8; ++j) { part[0][j] += m[j]; part[1][j] += m[j] * m[j]; }
    }
  }
  if (stats) block_channel_reduce<2>(part, oc, active, CP, stats, red);
}
static int maxpool2_fwd_impl(const void* x, void* y, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W,
                             int32_t CP, double* stats, SpQ8 q8, sp_stream_t stream, int64_t x8_plane = 0) {
  SP_CHECK_ARG(x && (y || q8.p) && CP % 8 == 0 && D >= 2 && H >= 2 && W >= 2, "sp_maxpool2_fwd: bad arguments");
  SP_CHECK_ARG(x8_plane == 0 || (dtype == SP_BF16 && CP % 16 == 0 && x8_plane >= (int64_t)B * D * H * W * 16), "sp_maxpool2_fwd_x8: e4m3 input planes of >= B D H W 16 bytes");
  SP_CHECK_ARG(!q8.p || (dtype == SP_BF16 && CP % 16 == 0 && q8.plane >= (int64_t)B * (D / 2) * (H / 2) * (W / 2) * 16 && q8.scale > 0.f),
               "sp_maxpool2_fwd_q8: bf16 tensors of whole 16-channel planes");
  SP_CHECK_VOX((int64_t)B * D * H * W, "sp_maxpool2_fwd");
  OctMap om = make_octmap(CP);
  Dims di{B, D, H, W};
  const int64_t nout = (int64_t)B * (D / 2) * (H / 2) * (W / 2);
  const unsigned grid = grid_for(nout, om.vpb * 2);
  const size_t sh = (size_t)CP * 2 * sizeof(float);
  if (dtype == SP_BF16) hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)x, (bf16_t*)y, di, CP, om, stats, q8, (int64_t)0, (int64_t)0, x8_plane);
  else hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid), dim3(256), sh, ST(stream), (const float*)x, (float*)y, di, CP, om, stats, q8, (int64_t)0, (int64_t)0, (int64_t)0);
  SP_CHECK_LAUNCH("sp_maxpool2_fwd");
  return SP_OK;
}
// the input as its e4m3 plane-major copy (the fp8 mode: the producing convolution stored no 16-bit tensor); y may be NULL (only the
// e4m3 copy of the result is wanted)
extern "C" int sp_maxpool2_fwd_x8(const void* x8, int64_t x8_plane, void* y, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP,
                                  double* stats, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(x8_plane > 0 && (!q8 || q8_fmt == 0 || q8_fmt == 1), "sp_maxpool2_fwd_x8: bad arguments");
  return maxpool2_fwd_impl(x8, y, SP_BF16, B, D, H, W, CP, stats, SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8 ? q8_scale : 1.f, q8_fmt}, stream, x8_plane);
}
// bf16 pairs (SP_HL): x / y are the hi halves, the lo halves x_lo_delta / y_lo_delta bytes behind them; the maximum of the pair
// VALUES (hi + lo) is written as a pair again, statistics of those values
extern "C" int sp_maxpool2_fwd_hl(const void* x, int64_t x_lo_delta, void* y, int64_t y_lo_delta, int32_t B, int32_t D, int32_t H, int32_t W,
                                  int32_t CP, double* stats, sp_stream_t stream) {
  SP_CHECK_ARG(x && y && x_lo_delta && y_lo_delta && x_lo_delta % 16 == 0 && y_lo_delta % 16 == 0 && CP % 8 == 0 && D >= 2 && H >= 2 && W >= 2, "sp_maxpool2_fwd_hl: bad arguments");
  SP_CHECK_VOX((int64_t)B * D * H * W, "sp_maxpool2_fwd_hl");
  OctMap om = make_octmap(CP);
  Dims di{B, D, H, W};
  const int64_t nout = (int64_t)B * (D / 2) * (H / 2) * (W / 2);
  const unsigned grid = grid_for(nout, om.vpb * 2);
  const size_t sh = (size_t)CP * 2 * sizeof(float);
  hipLaunchKernelGGL(maxpool2_fwd_kernel<sp_hl_t>, dim3(grid), dim3(256), sh, ST(stream), (const sp_hl_t*)x, (sp_hl_t*)y, di, CP, om, stats,
                     SpQ8{nullptr, 0, 1.f, 0}, x_lo_delta, y_lo_delta);
  SP_CHECK_LAUNCH("sp_maxpool2_fwd_hl");
  return SP_OK;
}
extern "C" int sp_maxpool2_fwd(const void* x, void* y, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W,
                               int32_t CP, double* stats, sp_stream_t stream) {
  return maxpool2_fwd_impl(x, y, dtype, B, D, H, W, CP, stats, SpQ8{nullptr, 0, 1.f, 0}, stream);
}
extern "C" int sp_maxpool2_fwd_q8(const void* x, void* y, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP,
                                  double* stats, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(q8 && (q8_fmt == 0 || q8_fmt == 1), "sp_maxpool2_fwd_q8: bad fp8 output");
  return maxpool2_fwd_impl(x, y, dtype, B, D, H, W, CP, stats, SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8_scale, q8_fmt}, stream);
}

// nn.Upsample(scale_factor=2, mode='trilinear'), align_corners=False (Unet3D.py:44,46):
// src = max(0, o/2 - 0.25); i0 = floor(src); i1 = min(i0+1, N-1); lambda = src - i0
__device__ __forceinline__ void up_src(int o, int N, int& i0, int& i1, float& l1) {
  float s = 0.5f * (float)o - 0.25f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + 1 < N ? i0 + 1 : N - 1;
  l1 = s - (float)i0;
}
template <typename T>
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, Dims di, int CP,
                                                             int CPd, OctMap om, double* __restrict__ stats) {
  extern __shared__ float red[];
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  const int Do = di.D * 2, Ho = di.H * 2, Wo = di.W * 2;
  const int64_t nout = (int64_t)di.B * Do * Ho * Wo;
  float part[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = part[1][j] = 0.f;
  const Unflat uf_(Do, Ho, Wo);
  if (active) {
    // each workgroup walks ONE contiguous voxel range (neighbouring rows stay in its L1 / the XCD's L2)
    const int64_t chunk_ = ((nout + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nout, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      int b, z, yy, xx;
      uf_(v, b, z, yy, xx);
      int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
      up_src(z, di.D, z0, z1, lz); up_src(yy, di.H, y0, y1, ly); up_src(xx, di.W, x0, x1, lx);
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = (k & 4) ? z1 : z0, iy = (k & 2) ? y1 : y0, ix = (k & 1) ? x1 : x0;
        const float w = ((k & 4) ? lz : 1.f - lz) * ((k & 2) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
        float a[8];
        Store<T>::ld8(x + ((((int64_t)b * di.D + iz) * di.H + iy) * di.W + ix) * CP + oc * 8, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(w, a[j], o[j]);
      }
      Store<T>::st8(y + v * CPd + oc * 8, o);
      if (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = bf2f(f2bf(o[j]));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { part[0][j] += o[j]; part[1][j] += o[j] * o[j]; }
    }
  }
  if (stats) block_channel_reduce<2>(part, oc, active, CP, stats, red);
}
extern "C" int sp_upsample2_fwd(const void* x, void* y, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W,
                                int32_t CP, int32_t CPd, double* stats, sp_stream_t stream) {
  SP_CHECK_ARG(x && y && CP % 8 == 0 && CPd >= CP && CPd % 8 == 0, "sp_upsample2_fwd: bad arguments");
  SP_CHECK_VOX((int64_t)B * D * H * W * 8, "sp_upsample2_fwd");
  OctMap om = make_octmap(CP);
  Dims di{B, D, H, W};
  const int64_t nout = (int64_t)B * D * H * W * 8;
  const unsigned grid = grid_for(nout, om.vpb * 2);
  const size_t sh = (size_t)CP * 2 * sizeof(float);
  if (dtype == SP_BF16) hipLaunchKernelGGL(upsample2_fwd_kernel<bf16_t>, dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)x, (bf16_t*)y, di, CP, CPd, om, stats);
  else hipLaunchKernelGGL(upsample2_fwd_kernel<float>, dim3(grid), dim3(256), sh, ST(stream), (const float*)x, (float*)y, di, CP, CPd, om, stats);
  SP_CHECK_LAUNCH("sp_upsample2_fwd");
  return SP_OK;
}

// centre crop (offset (in-out)//2, Unet3D.py:10) of src into channels [c0, c0+CPs) of dst
template <typename T>
__global__ __launch_bounds__(256) void crop_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, Dims ds, int CPs,
                                                         Dims dd, int CPd, int c0, OctMap om, double* __restrict__ stats) {
  extern __shared__ float red[];
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  const int oz = (ds.D - dd.D) / 2, oy = (ds.H - dd.H) / 2, ox = (ds.W - dd.W) / 2;
  const int64_t nout = (int64_t)dd.B * dd.D * dd.H * dd.W;
  float part[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = part[1][j] = 0.f;
  const Unflat uf_(dd.D, dd.H, dd.W);
  if (active) {
    // each workgroup walks ONE contiguous voxel range (neighbouring rows stay in its L1 / the XCD's L2)
    const int64_t chunk_ = ((nout + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nout, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      int b, z, yy, xx;
      uf_(v, b, z, yy, xx);
      float a[8];
      Store<T>::ld8(src + ((((int64_t)b * ds.D + z + oz) * ds.H + yy + oy) * ds.W + xx + ox) * CPs + oc * 8, a);
      Store<T>::st8(dst + v * CPd + c0 + oc * 8, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) { part[0][j] += a[j]; part[1][j] += a[j] * a[j]; }
    }
  }
  if (stats) block_channel_reduce<2>(part, oc, active, CPs, stats, red);
}
extern "C" int sp_crop_copy(const void* src, void* dst, int32_t dtype, int32_t B, int32_t Ds, int32_t Hs, int32_t Ws,
                            int32_t CPs, int32_t Dd, int32_t Hd, int32_t Wd, int32_t CPd, int32_t c0, double* stats,
                            sp_stream_t stream) {
  SP_CHECK_ARG(src && dst && CPs % 8 == 0 && CPd % 8 == 0 && c0 % 8 == 0 && c0 + CPs <= CPd, "sp_crop_copy: bad channels");
  SP_CHECK_VOX((int64_t)B * Ds * Hs * Ws, "sp_crop_copy");
  SP_CHECK_ARG(Dd <= Ds && Hd <= Hs && Wd <= Ws, "sp_crop_copy: crop larger than source");
  OctMap om = make_octmap(CPs);
  Dims ds{B, Ds, Hs, Ws}, dd{B, Dd, Hd, Wd};
  const unsigned grid = grid_for((int64_t)B * Dd * Hd * Wd, om.vpb * 4);
  const size_t sh = (size_t)CPs * 2 * sizeof(float);
  if (dtype == SP_BF16) hipLaunchKernelGGL(crop_copy_kernel<bf16_t>, dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)src, (bf16_t*)dst, ds, CPs, dd, CPd, c0, om, stats);
  else hipLaunchKernelGGL(crop_copy_kernel<float>, dim3(grid), dim3(256), sh, ST(stream), (const float*)src, (float*)dst, ds, CPs, dd, CPd, c0, om, stats);
  SP_CHECK_LAUNCH("sp_crop_copy");
  return SP_OK;
}

// eight elements kept as loaded (bf16: 4 VGPRs instead of 8) and unpacked where used
template <typename T> struct RawOct;
template <> struct RawOct<bf16_t> {
  uint4 r;
  __device__ __forceinline__ void load(const bf16_t* p, int64_t = 0) { r = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float* v) const {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = sp_h2f_lo(w[i]); v[2 * i + 1] = sp_h2f_hi(w[i]); }
  }
};
template <> struct RawOct<float> {
  float f[8];
  __device__ __forceinline__ void load(const float* p, int64_t = 0) { Store<float>::ld8(p, f); }
  __device__ __forceinline__ void get(float* v) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f[i];
  }
};
template <> struct RawOct<sp_hl_t> {      // bf16 pair: hi and lo words as loaded
  uint4 h, l;
  __device__ __forceinline__ void load(const sp_hl_t* p, int64_t lo_delta) {
    h = *reinterpret_cast<const uint4*>(p);
    l = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p) + lo_delta);
  }
};

// ---- packed fp32 pairs: v_pk_fma_f32 / v_pk_mul_f32 do two fp32 operations per lane and instruction on gfx950; the
// gather kernels below are VALU-bound (the 4-channels-per-thread variant of upcat_fwd, 126 instead of 213 VGPRs and twice
// the occupancy, was 10 % SLOWER), so their interpolation arithmetic runs on channel pairs
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2_t f2_fma(f2_t a, f2_t b, f2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2_t f2_splat(float v) { f2_t r = {v, v}; return r; }
__device__ __forceinline__ void raw_get2(const RawOct<bf16_t>& r, f2_t* v) {
  const uint32_t w[4] = {r.r.x, r.r.y, r.r.z, r.r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f2_t t = {sp_h2f_lo(w[i]), sp_h2f_hi(w[i])}; v[i] = t; }
}
__device__ __forceinline__ void raw_get2(const RawOct<float>& r, f2_t* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f2_t t = {r.f[2 * i], r.f[2 * i + 1]}; v[i] = t; }
}
__device__ __forceinline__ void raw_get2(const RawOct<sp_hl_t>& r, f2_t* v) {
  const uint32_t hw[4] = {r.h.x, r.h.y, r.h.z, r.h.w}, lw[4] = {r.l.x, r.l.y, r.l.z, r.l.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f2_t t = {sp_h2f_lo(hw[i]) + sp_h2f_lo(lw[i]), sp_h2f_hi(hw[i]) + sp_h2f_hi(lw[i])}; v[i] = t; }
}
__device__ __forceinline__ void st8_f2(bf16_t* p, const f2_t* v) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = sp_pack_bf16x2(v[i].x, v[i].y);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void st8_f2(float* p, const f2_t* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
}
// bf16 pair: split and store both halves (the values stay what they are: the statistics see the fp32 values)
__device__ __forceinline__ void st8_f2_hl(sp_hl_t* p, int64_t lo_delta, const f2_t* v) {
  const float f[8] = {v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y};
  sp_hl_st8(p, lo_delta, f);
}
__device__ __forceinline__ f2_t round_like(const sp_hl_t*, f2_t v) { return v; }
__device__ __forceinline__ f2_t round_like(const bf16_t*, f2_t v) {      // one packed conversion, two bit operations
  const uint32_t u = sp_pack_bf16x2(v.x, v.y);
  f2_t r = {sp_h2f_lo(u), sp_h2f_hi(u)};
  return r;
}
// store eight values and return them as stored (what the statistics must see): bf16 converts each pair ONCE
__device__ __forceinline__ void st8_f2_rounded(bf16_t* p, f2_t* v) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w[i] = sp_pack_bf16x2(v[i].x, v[i].y);
    f2_t r = {sp_h2f_lo(w[i]), sp_h2f_hi(w[i])};
    v[i] = r;
  }
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void st8_f2_rounded(float* p, f2_t* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
}
__device__ __forceinline__ f2_t round_like(const float*, f2_t v) { return v; }

// Upsample + crop + concat in ONE pass (Unet3D.py:67-72): every voxel row of the concat buffer (all CPd channels) is
// written by neighbouring lanes -- full lines -- instead of two kernels each writing a slice of every row.  Channels
// [0, CPu) are the trilinear x2 upsample of `low`, channels [CPu, CPu + CPs) the centre crop of `skip`;
// stats[c][2] += (sum, sum^2).
// One thread = one 2x2x2 OUTPUT block x 8 channels.  The block's eight upsampled voxels depend on the 3x3x3 source
// neighbourhood (i-1, i, i+1 per axis, clamped): out(2i) = .25 s(i-1) + .75 s(i), out(2i+1) = .75 s(i) + .25 s(i+1),
// evaluated separably (x, then y, then z) -- 27 loads and ~80 vector FMAs for eight outputs instead of 64 and 64
// (the one-voxel-per-thread version was ALU-bound: ~400 instructions per output octet).
template <typename T>
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const T* __restrict__ low, Dims dl, int CPu, const T* __restrict__ skip,
                                                         Dims ds, int CPs, T* __restrict__ cat, int CPd, int64_t cat_plane, OctMap om,
                                                         double* __restrict__ stats, int64_t low_lo = 0, int64_t skip_lo = 0, int64_t cat_lo = 0) {
  extern __shared__ float red[];
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  const int Do = 2 * dl.D, Ho = 2 * dl.H, Wo = 2 * dl.W;
  const int oz = (ds.D - Do) / 2, oy = (ds.H - Ho) / 2, ox = (ds.W - Wo) / 2;
  const int ocu = CPu / 8;
  const bool isup = oc < ocu;
  const int64_t nblk = (int64_t)dl.B * dl.D * dl.H * dl.W;       // one block per source voxel
  f2_t part2[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) part2[0][j] = part2[1][j] = f2_splat(0.f);
  const Unflat uf_(dl.D, dl.H, dl.W);
  if (active) {
    const int64_t chunk_ = ((nblk + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nblk, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      int b, z, yy, xx;
      uf_(v, b, z, yy, xx);
      f2_t out[8][4];                                              // [zo*4 + yo*2 + xo][channel pair]
      if (isup) {
        const int xs[3] = {max(xx - 1, 0), xx, min(xx + 1, dl.W - 1)};
        const int ys[3] = {max(yy - 1, 0), yy, min(yy + 1, dl.H - 1)};
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) out[q][j] = f2_splat(0.f);
#pragma unroll 1
        for (int dz = 0; dz < 3; ++dz) {
          const int zs = dz == 0 ? max(z - 1, 0) : (dz == 1 ? z : min(z + 1, dl.D - 1));
          const f2_t wz0 = f2_splat(dz == 0 ? 0.25f : (dz == 1 ? 0.75f : 0.f));      // weight into output plane 2z
          const f2_t wz1 = f2_splat(dz == 0 ? 0.f : (dz == 1 ? 0.75f : 0.25f));      // ... and 2z + 1
          const T* pz = low + (((int64_t)b * dl.D + zs) * dl.H) * dl.W * CPu + oc * 8;
          // x-interpolated rows (even / odd output x) of the three source rows, then the two output y
          f2_t xe[3][4], xo[3][4];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            RawOct<T> r0, r1, r2;
            const T* py = pz + (int64_t)ys[dy] * dl.W * CPu;
            r0.load(py + (int64_t)xs[0] * CPu, low_lo); r1.load(py + (int64_t)xs[1] * CPu, low_lo); r2.load(py + (int64_t)xs[2] * CPu, low_lo);
            f2_t a0[4], a1[4], a2[4];
            raw_get2(r0, a0); raw_get2(r1, a1); raw_get2(r2, a2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f2_t m = a1[j] * 0.75f;
              xe[dy][j] = f2_fma(f2_splat(0.25f), a0[j], m);
              xo[dy][j] = f2_fma(f2_splat(0.25f), a2[j], m);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f2_t me = xe[1][j] * 0.75f, mo = xo[1][j] * 0.75f;
            const f2_t ye0 = f2_fma(f2_splat(0.25f), xe[0][j], me), ye1 = f2_fma(f2_splat(0.25f), xo[0][j], mo);   // even output y
            const f2_t yo0 = f2_fma(f2_splat(0.25f), xe[2][j], me), yo1 = f2_fma(f2_splat(0.25f), xo[2][j], mo);   // odd output y
            out[0][j] = f2_fma(wz0, ye0, out[0][j]); out[1][j] = f2_fma(wz0, ye1, out[1][j]);
            out[2][j] = f2_fma(wz0, yo0, out[2][j]); out[3][j] = f2_fma(wz0, yo1, out[3][j]);
            out[4][j] = f2_fma(wz1, ye0, out[4][j]); out[5][j] = f2_fma(wz1, ye1, out[5][j]);
            out[6][j] = f2_fma(wz1, yo0, out[6][j]); out[7][j] = f2_fma(wz1, yo1, out[7][j]);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int zo = 2 * z + (q >> 2), yo = 2 * yy + ((q >> 1) & 1), xo = 2 * xx + (q & 1);
          RawOct<T> r;
          r.load(skip + ((((int64_t)b * ds.D + zo + oz) * ds.H + yo + oy) * ds.W + xo + ox) * CPs + (oc - ocu) * 8, skip_lo);
          raw_get2(r, out[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int zo = 2 * z + (q >> 2), yo = 2 * yy + ((q >> 1) & 1), xo = 2 * xx + (q & 1);
        const int64_t vo = (((int64_t)b * Do + zo) * Ho + yo) * Wo + xo;
        // cat_plane != 0: plane-major concat buffer [plane][B][D][H][W][16] (dense 16-channel planes for the consumers)
        T* const dst_ = cat_plane ? cat + (int64_t)(oc >> 1) * cat_plane + vo * 16 + (oc & 1) * 8 : cat + vo * CPd + oc * 8;
        if constexpr (std::is_same<T, sp_hl_t>::value) st8_f2_hl(dst_, cat_lo, out[q]); else st8_f2(dst_, out[q]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2_t o = round_like(cat, out[q][j]);
          part2[0][j] += o; part2[1][j] = f2_fma(o, o, part2[1][j]);
        }
      }
    }
  }
  float part[2][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    part[0][2 * j] = part2[0][j].x; part[0][2 * j + 1] = part2[0][j].y;
    part[1][2 * j] = part2[1][j].x; part[1][2 * j + 1] = part2[1][j].y;
  }
  if (stats) block_channel_reduce<2>(part, oc, active, CPd, stats, red);
}
// ---- plane-major output, row-ordered variant.  The block-per-thread kernel above writes 16-byte pieces 64 bytes apart
// (a lane's eight voxels are its own 2x2x2 block; neighbouring lanes are other channel octets = other planes): 140 us for
// the 424 MB of the last concat, 3 TB/s.  Here blockIdx.y is the 16-channel plane and consecutive lanes are consecutive
// (output x, channel half) pairs, so every store instruction of a wave writes 1 KB of one output row; a thread produces
// the four outputs (2 z x 2 y) above its output x from 3 x 3 x 2 source octets (neighbouring lanes share them in L1).
template <typename T>
__global__ __launch_bounds__(256) void upcat_rows_kernel(const T* __restrict__ low, Dims dl, int CPu, const T* __restrict__ skip,
                                                          Dims ds, int CPs, T* __restrict__ cat, int CPd, int64_t cat_plane,
                                                          double* __restrict__ stats, const SpQ8 q8,
                                                          int64_t low_lo = 0, int64_t skip_lo = 0, int64_t cat_lo = 0, int64_t skip8_plane = 0) {
  // skip8_plane > 0: skip is the e4m3 plane-major copy [CPs/16][B][Ds][Hs][Ws][16 bytes] of the skip tensor (fp8 mode: no 16-bit one)
  __shared__ float red[4 * 32];
  // 1-D grid of gx * np workgroups, gx a multiple of 8: workgroup id -> (chunk, plane) such that the np planes of a chunk are
  // CONSECUTIVE workgroups of ONE XCD (ids 8 apart share an L2).  A plane takes 32 bytes of every source voxel; with the planes
  // of a chunk far apart in time (plane = blockIdx.y) every 64-byte sector of the skip tensor and every 128-byte line of the low
  // tensor came in once per plane.
  const int np = CPd >> 4, nup = CPu >> 4;
  const uint32_t gx = gridDim.x / np, t = blockIdx.x >> 3;
  const int p = t % np;
  const uint32_t cx = (t / np) * 8 + (blockIdx.x & 7);
  const int Do = 2 * dl.D, Ho = 2 * dl.H, Wo = 2 * dl.W;
  const int oz = (ds.D - Do) / 2, oy = (ds.H - Ho) / 2, ox = (ds.W - Wo) / 2;
  const int64_t total = (int64_t)dl.B * dl.D * dl.H * Wo * 2;
  const int64_t chunk = ((total + gx - 1) / gx + 255) / 256 * 256;
  const int64_t i0 = (int64_t)cx * chunk, i1 = min(total, i0 + chunk);
  const int half = threadIdx.x & 1;                   // chunk and stride are even: a thread keeps its channel half
  const FastDiv d_w2 = make_fastdiv(Wo * 2), d_h = make_fastdiv(dl.H), d_d = make_fastdiv(dl.D);
  f2_t s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) s1[j] = s2[j] = f2_splat(0.f);
  T* const cp = cat ? cat + (int64_t)p * cat_plane + half * 8 : nullptr;      // (NULL: fp8 copy and statistics only)
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    uint32_t r = fdiv((uint32_t)i, d_w2);
    const int xo = ((uint32_t)i - r * (Wo * 2)) >> 1;
    uint32_t q = fdiv(r, d_h); const int yl = r - q * dl.H;
    r = fdiv(q, d_d); const int zl = q - r * dl.D; const int b = r;
    f2_t out[4][4];                                     // [zo*2 + yo][channel pair]
    if (p < nup) {
      const int xi = xo >> 1, odd = xo & 1;
      const int xa = max(odd ? xi : xi - 1, 0), xb = min(odd ? xi + 1 : xi, dl.W - 1);
      const f2_t wa = f2_splat(odd ? 0.75f : 0.25f), wb = f2_splat(odd ? 0.25f : 0.75f);
      const int ys[3] = {max(yl - 1, 0), yl, min(yl + 1, dl.H - 1)};
      const int zs[3] = {max(zl - 1, 0), zl, min(zl + 1, dl.D - 1)};
      const T* base = low + p * 16 + half * 8;
      f2_t ye[3][4], yo[3][4];                          // per source plane: even / odd output row
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        f2_t v[3][4];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const T* row = base + (((int64_t)b * dl.D + zs[dz]) * dl.H + ys[dy]) * dl.W * CPu;
          RawOct<T> ra, rb;
          ra.load(row + (int64_t)xa * CPu, low_lo); rb.load(row + (int64_t)xb * CPu, low_lo);
          f2_t a[4], c[4];
          raw_get2(ra, a); raw_get2(rb, c);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[dy][j] = f2_fma(wa, a[j], wb * c[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2_t m = v[1][j] * 0.75f;
          ye[dz][j] = f2_fma(f2_splat(0.25f), v[0][j], m);
          yo[dz][j] = f2_fma(f2_splat(0.25f), v[2][j], m);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f2_t me = ye[1][j] * 0.75f, mo = yo[1][j] * 0.75f;
        out[0][j] = f2_fma(f2_splat(0.25f), ye[0][j], me); out[1][j] = f2_fma(f2_splat(0.25f), yo[0][j], mo);
        out[2][j] = f2_fma(f2_splat(0.25f), ye[2][j], me); out[3][j] = f2_fma(f2_splat(0.25f), yo[2][j], mo);
      }
    } else {
      const T* base = skip + (p - nup) * 16 + half * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t vs_ = (((int64_t)b * ds.D + 2 * zl + (k >> 1) + oz) * ds.H + 2 * yl + (k & 1) + oy) * ds.W + xo + ox;
        if (skip8_plane) {
          float f8_[8];
          sp_ld8_e4m3(skip, skip8_plane, vs_, 2 * (p - nup) + half, f8_);
#pragma unroll
          for (int j = 0; j < 4; ++j) { out[k][j].x = f8_[2 * j]; out[k][j].y = f8_[2 * j + 1]; }
        } else {
          RawOct<T> rr;
          rr.load(base + vs_ * CPs, skip_lo);
          raw_get2(rr, out[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t vo = (((int64_t)b * Do + 2 * zl + (k >> 1)) * Ho + 2 * yl + (k & 1)) * Wo + xo;
      if constexpr (std::is_same<T, sp_hl_t>::value) { if (cp) st8_f2_hl(cp + vo * 16, cat_lo, out[k]); }
      else if (cp) st8_f2_rounded(cp + vo * 16, out[k]);
      if (q8.p) {
        float o8[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { o8[2 * j] = out[k][j].x; o8[2 * j + 1] = out[k][j].y; }
        sp_q8_store8(q8, vo, 2 * p + half, o8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] += out[k][j]; s2[j] = f2_fma(out[k][j], out[k][j], s2[j]); }
    }
  }
  if (stats) {       // channel c = p*16 + half*8 + 2j + {0,1}: lanes of one parity hold the same eight channels
    __syncthreads();      // (red: [4 waves][32], added up in wave order -- sp_cols_sum)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v4[4] = {s1[j].x, s2[j].x, s1[j].y, s2[j].y};         // (sum, sum^2) of channel 2j, then of channel 2j + 1
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = v4[k];
#pragma unroll
        for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) < 2) red[(threadIdx.x >> 6) * 32 + (half * 8 + 2 * j + (k >> 1)) * 2 + (k & 1)] = v;
      }
    }
    __syncthreads();
    if (threadIdx.x < 32)
      atomicAdd(&stats[(size_t)(cx % SP_REDUCE_ROWS) * CPd * 2 + (size_t)p * 32 + threadIdx.x], (double)sp_cols_sum(red, 32, 4, threadIdx.x));
  }
}

static int upcat_impl(const void* low, int32_t CPu, const void* skip, int32_t CPs, void* cat, int32_t CPd,
                      int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds, int32_t Hs,
                      int32_t Ws, int64_t cat_plane, double* stats, SpQ8 q8, sp_stream_t stream, const int64_t* hl_lo = nullptr,
                      int64_t skip8_plane = 0) {
  static const int64_t no_lo_[3] = {0, 0, 0};
  SP_CHECK_ARG(dtype != SP_HL || (hl_lo && hl_lo[0] && hl_lo[1] && hl_lo[2] && !q8.p && cat), "sp_upsample2_crop_cat_fwd_hl: bf16 pairs need the three lo deltas");
  if (!hl_lo) hl_lo = no_lo_;
  // cat == NULL: only the fp8 copy (and the statistics) are wanted -- rows kernel only
  SP_CHECK_ARG(low && skip && (cat || q8.p) && CPu % 8 == 0 && CPs % 8 == 0 && CPd == CPu + CPs, "sp_upsample2_crop_cat_fwd: bad channels");
  SP_CHECK_ARG(cat || (cat_plane && CPu % 16 == 0 && CPs % 16 == 0 && (int64_t)B * D * H * W * 4 < (1ll << 31) && !getenv("SP_UPCAT_BLOCKS")),
               "sp_upsample2_crop_cat_fwd_q8: without the 16-bit output the plane-major form is required");
  SP_CHECK_VOX((int64_t)B * Ds * Hs * Ws, "sp_upsample2_crop_cat_fwd");
  SP_CHECK_ARG(2 * D <= Ds && 2 * H <= Hs && 2 * W <= Ws, "sp_upsample2_crop_cat_fwd: skip smaller than the upsampled grid");
  SP_CHECK_ARG(CPd <= 2048, "sp_upsample2_crop_cat_fwd: too many channels");
  SP_CHECK_ARG(cat_plane == 0 || CPd % 16 == 0, "sp_upsample2_crop_cat_fwd: plane-major output needs whole 16-channel planes");
  OctMap om = make_octmap(CPd);
  Dims dl{B, D, H, W}, ds{B, Ds, Hs, Ws};
  if (cat_plane && CPu % 16 == 0 && CPs % 16 == 0 && (int64_t)B * D * H * W * 4 < (1ll << 31) && !getenv("SP_UPCAT_BLOCKS")) {
    const int64_t total = (int64_t)B * D * H * W * 4;     // (output x, channel half) pairs over the source rows
    // workgroups over all planes: 2048 left 2064 workgroups for 1024 resident ones (a third, nearly empty round); swept 2048 .. 65536
    // at 2 x 168^3 x 96 channels: 752 / 664 / 630 / 643 / 642 us (SP_UPCAT_CAP: the sweep's knob)
    static const int cap_total_ = getenv("SP_UPCAT_CAP") ? atoi(getenv("SP_UPCAT_CAP")) : 8192;
    SP_CHECK_ARG(cap_total_ >= 8, "sp_upsample2_crop_cat_fwd: SP_UPCAT_CAP %d", cap_total_);
    const int64_t want = (total + 1023) / 1024, cap = cap_total_ / (CPd / 16) + 1;
    const unsigned gx = ((unsigned)(want < cap ? want : cap) + 7) / 8 * 8;
    dim3 grid(gx * (unsigned)(CPd / 16));
    SP_CHECK_ARG(!q8.p || (dtype == SP_BF16 && q8.plane >= (int64_t)B * D * H * W * 8 * 16 && q8.scale > 0.f), "sp_upsample2_crop_cat_fwd_q8: bf16 tensors only");
    SP_CHECK_ARG(skip8_plane == 0 || (dtype == SP_BF16 && skip8_plane >= (int64_t)B * Ds * Hs * Ws * 16), "sp_upsample2_crop_cat_fwd_q8s8: e4m3 skip planes of >= B Ds Hs Ws 16 bytes");
    if (dtype == SP_BF16) hipLaunchKernelGGL(upcat_rows_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), (const bf16_t*)low, dl, CPu, (const bf16_t*)skip, ds, CPs, (bf16_t*)cat, CPd, cat_plane, stats, q8, (int64_t)0, (int64_t)0, (int64_t)0, skip8_plane);
    else if (dtype == SP_HL) hipLaunchKernelGGL(upcat_rows_kernel<sp_hl_t>, grid, dim3(256), 0, ST(stream), (const sp_hl_t*)low, dl, CPu, (const sp_hl_t*)skip, ds, CPs, (sp_hl_t*)cat, CPd, cat_plane, stats, q8, hl_lo[0], hl_lo[1], hl_lo[2]);
    else hipLaunchKernelGGL(upcat_rows_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)low, dl, CPu, (const float*)skip, ds, CPs, (float*)cat, CPd, cat_plane, stats, q8, (int64_t)0, (int64_t)0, (int64_t)0);
    SP_CHECK_LAUNCH("sp_upsample2_crop_cat_fwd(rows)");
    return SP_OK;
  }
  SP_CHECK_ARG(!q8.p && !skip8_plane, "sp_upsample2_crop_cat_fwd_q8: the fp8 copy is written (and an e4m3 skip tensor read) by the plane-major (row-ordered) kernel only: cat_plane != 0, channel counts multiples of 16");
  const int64_t nblk = (int64_t)B * D * H * W;          // one thread-slot per 2x2x2 output block
  const unsigned grid = grid_for(nblk, om.vpb);
  const size_t sh = (size_t)CPd * 2 * sizeof(float);
  if (dtype == SP_BF16) hipLaunchKernelGGL(upcat_fwd_kernel<bf16_t>, dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)low, dl, CPu, (const bf16_t*)skip, ds, CPs, (bf16_t*)cat, CPd, cat_plane, om, stats, (int64_t)0, (int64_t)0, (int64_t)0);
  else if (dtype == SP_HL) hipLaunchKernelGGL(upcat_fwd_kernel<sp_hl_t>, dim3(grid), dim3(256), sh, ST(stream), (const sp_hl_t*)low, dl, CPu, (const sp_hl_t*)skip, ds, CPs, (sp_hl_t*)cat, CPd, cat_plane, om, stats, hl_lo[0], hl_lo[1], hl_lo[2]);
  else hipLaunchKernelGGL(upcat_fwd_kernel<float>, dim3(grid), dim3(256), sh, ST(stream), (const float*)low, dl, CPu, (const float*)skip, ds, CPs, (float*)cat, CPd, cat_plane, om, stats, (int64_t)0, (int64_t)0, (int64_t)0);
  SP_CHECK_LAUNCH("sp_upsample2_crop_cat_fwd");
  return SP_OK;
}
extern "C" int sp_upsample2_crop_cat_fwd(const void* low, int32_t CPu, const void* skip, int32_t CPs, void* cat, int32_t CPd,
                                         int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds, int32_t Hs,
                                         int32_t Ws, int64_t cat_plane, double* stats, sp_stream_t stream) {
  return upcat_impl(low, CPu, skip, CPs, cat, CPd, dtype, B, D, H, W, Ds, Hs, Ws, cat_plane, stats, SpQ8{nullptr, 0, 1.f, 0}, stream);
}
// bf16 pairs (SP_HL): low / skip / cat are the hi halves of pair tensors, their lo halves *_lo_delta bytes behind them; the
// interpolation runs on the pair values in fp32 and the result is split again (cat_plane: in ELEMENTS of one half, as above)
extern "C" int sp_upsample2_crop_cat_fwd_hl(const void* low, int64_t low_lo_delta, int32_t CPu, const void* skip, int64_t skip_lo_delta, int32_t CPs,
                                            void* cat, int64_t cat_lo_delta, int32_t CPd, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds,
                                            int32_t Hs, int32_t Ws, int64_t cat_plane, double* stats, sp_stream_t stream) {
  const int64_t lo[3] = {low_lo_delta, skip_lo_delta, cat_lo_delta};
  SP_CHECK_ARG(low_lo_delta % 16 == 0 && skip_lo_delta % 16 == 0 && cat_lo_delta % 16 == 0, "sp_upsample2_crop_cat_fwd_hl: lo halves must be 16-byte aligned");
  return upcat_impl(low, CPu, skip, CPs, cat, CPd, SP_HL, B, D, H, W, Ds, Hs, Ws, cat_plane, stats, SpQ8{nullptr, 0, 1.f, 0}, stream, lo);
}
extern "C" int sp_upsample2_crop_cat_fwd_q8(const void* low, int32_t CPu, const void* skip, int32_t CPs, void* cat, int32_t CPd,
                                            int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds, int32_t Hs,
                                            int32_t Ws, int64_t cat_plane, double* stats, void* q8, int64_t q8_plane, int32_t q8_fmt,
                                            float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(q8 && (q8_fmt == 0 || q8_fmt == 1), "sp_upsample2_crop_cat_fwd_q8: bad fp8 output");
  return upcat_impl(low, CPu, skip, CPs, cat, CPd, dtype, B, D, H, W, Ds, Hs, Ws, cat_plane, stats,
                    SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8_scale, q8_fmt}, stream);
}
// the skip tensor as its e4m3 plane-major copy (fp8 mode: the producing convolution stored no 16-bit tensor); low stays 16-bit
extern "C" int sp_upsample2_crop_cat_fwd_q8s8(const void* low, int32_t CPu, const void* skip8, int64_t skip8_plane, int32_t CPs, void* cat,
                                              int32_t CPd, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds, int32_t Hs,
                                              int32_t Ws, int64_t cat_plane, double* stats, void* q8, int64_t q8_plane, int32_t q8_fmt,
                                              float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(q8 && (q8_fmt == 0 || q8_fmt == 1) && skip8_plane > 0 && cat_plane > 0, "sp_upsample2_crop_cat_fwd_q8s8: bad arguments (plane-major output only)");
  return upcat_impl(low, CPu, skip8, CPs, cat, CPd, SP_BF16, B, D, H, W, Ds, Hs, Ws, cat_plane, stats,
                    SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8_scale, q8_fmt}, stream, nullptr, skip8_plane);
}

// ------------------------------------------------------------------------------------------------ fused backward pieces
// Block output y feeds MaxPool3d(2,2) (-> next block's BN) and, centre-cropped, the skip concat.
// One thread = one 2x2x2 window x 8 channels: argmax is the FIRST maximum in (z,y,x) scan order
// (ATen max_pool3d), pool gradient = coefp0*gp + coefp1*p + coefp2 with p = max recomputed here.
template <typename T, int ACT, bool Y8 = false>      // Y8: y is its e4m3 plane-major copy (a compile-time switch: both register sets at once spill)
__global__ __launch_bounds__(256, 4) void pool_skip_act_bwd_kernel(
    const T* __restrict__ y, const T* __restrict__ gp, const float* __restrict__ coefp, const T* __restrict__ cat,
    const T* __restrict__ gs, const float* __restrict__ coefs, int cs0, int CPcat, int ccs0, int cstride, Dims di, int CP, Dims dc,
    OctMap om, int act, float ap, T* __restrict__ dz, double* __restrict__ dbias, const SpQ8 q8, int64_t y8_plane = 0) {
  // y8_plane > 0: y is its e4m3 plane-major copy (fp8 mode: no 16-bit tensor was stored)
  extern __shared__ float red[];
  float* cpl = red + CP;                              // pool-side coefficients [3][CP]: used once per window -> LDS, not VGPRs
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  const int Dw = (di.D + 1) / 2, Hw = (di.H + 1) / 2, Ww = (di.W + 1) / 2;      // windows incl. ragged edge
  const int Dp = di.D / 2, Hp = di.H / 2, Wp = di.W / 2;                        // pooled dims (floor)
  const int cz = (di.D - dc.D) / 2, cy = (di.H - dc.H) / 2, cx = (di.W - dc.W) / 2;
  const int64_t nwin = (int64_t)di.B * Dw * Hw * Ww;
  float* csl = red + 4 * CP;                          // skip-side coefficients [3][CP]: LDS too (24 VGPRs less: 150 -> <= 128, one more wave per SIMD)
  for (int i = threadIdx.x; i < 3 * CP; i += 256) {
    cpl[i] = gp ? coefp[i] : 0.f;
    const int r = i / CP, c = i - r * CP;
    csl[i] = gs ? coefs[r * cstride + ccs0 + c] : 0.f;
  }
  __syncthreads();
  float part[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = 0.f;
  const Unflat uf_(Dw, Hw, Ww);
  if (active) {
    // each workgroup walks ONE contiguous voxel range (neighbouring rows stay in its L1 / the XCD's L2)
    const int64_t chunk_ = ((nwin + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nwin, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      int b, wz, wy, wx;
      uf_(v, b, wz, wy, wx);
      const bool pooled = gp && wz < Dp && wy < Hp && wx < Wp;
      RawOct<T> yr[Y8 ? 1 : 8];
      uint2 yr8[Y8 ? 8 : 1];
      float m[8]; int am[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; am[j] = 0; }
      // the eight loads first (clamped coordinates: a branch around a load serialises it), then the scan
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = min(2 * wz + (k >> 2), di.D - 1), iy = min(2 * wy + ((k >> 1) & 1), di.H - 1), ix = min(2 * wx + (k & 1), di.W - 1);
        const int64_t vi_ = (((int64_t)b * di.D + iz) * di.H + iy) * di.W + ix;
        if constexpr (Y8) yr8[k] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(y) + (int64_t)(oc >> 1) * y8_plane + vi_ * 16 + (oc & 1) * 8);
        else yr[k].load(y + vi_ * CP + oc * 8);
      }
      auto get_y = [&](int k, float* out) {
        if constexpr (Y8) {
          typedef float f2v_ __attribute__((ext_vector_type(2)));
          const f2v_ a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yr8[k].x, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yr8[k].x, true);
          const f2v_ a2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yr8[k].y, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yr8[k].y, true);
          out[0] = a0[0]; out[1] = a0[1]; out[2] = a1[0]; out[3] = a1[1]; out[4] = a2[0]; out[5] = a2[1]; out[6] = a3[0]; out[7] = a3[1];
        } else yr[k].get(out);
      };
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = 2 * wz + (k >> 2), iy = 2 * wy + ((k >> 1) & 1), ix = 2 * wx + (k & 1);
        if (iz < di.D && iy < di.H && ix < di.W) {
          float yv[8];
          get_y(k, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) if (yv[j] > m[j]) { m[j] = yv[j]; am[j] = k; }
        }
      }
      float dp[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) dp[j] = 0.f;
      if (pooled) {
        float g8[8];
        Store<T>::ld8(gp + ((((int64_t)b * Dp + wz) * Hp + wy) * Wp + wx) * CP + oc * 8, g8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = oc * 8 + j;
          dp[j] = cpl[c] * g8[j] + cpl[CP + c] * m[j] + cpl[2 * CP + c];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int iz = 2 * wz + (k >> 2), iy = 2 * wy + ((k >> 1) & 1), ix = 2 * wx + (k & 1);
        if (iz < di.D && iy < di.H && ix < di.W) {
          float d[8], yv[8];
          get_y(k, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = (pooled && am[j] == k) ? dp[j] : 0.f;
          const int qz = iz - cz, qy = iy - cy, qx = ix - cx;
          if (gs && (unsigned)qz < (unsigned)dc.D && (unsigned)qy < (unsigned)dc.H && (unsigned)qx < (unsigned)dc.W) {
            const int64_t o = ((((int64_t)b * dc.D + qz) * dc.H + qy) * dc.W + qx) * CPcat + cs0 + oc * 8;
            float g8[8];
            Store<T>::ld8(gs + o, g8);
            // the skip half of the concat buffer is a verbatim crop of y: its value is the y just loaded
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int c = oc * 8 + j;
              d[j] += csl[c] * g8[j] + csl[CP + c] * yv[j] + csl[2 * CP + c];
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) { d[j] *= act_bwd_t<ACT>(act, ap, yv[j]); part[0][j] += d[j]; }
          const int64_t vo_ = (((int64_t)b * di.D + iz) * di.H + iy) * di.W + ix;
          if (dz) Store<T>::st8(dz + vo_ * CP + oc * 8, d);
          if (q8.p) sp_q8_store8(q8, vo_, oc, d);
        }
      }
    }
  }
  if (dbias) block_channel_reduce<1>(part, oc, active, CP, dbias, red);
}
static int pool_skip_impl(const void* y, const void* gp, const float* coefp, const void* cat, const void* gs,
                          const float* coefs, int32_t cs0, int32_t CPcat, int32_t coef_c0, int32_t coef_stride,
                          int32_t dtype, int32_t B, int32_t D,
                          int32_t H, int32_t W, int32_t CP, int32_t Dc, int32_t Hc, int32_t Wc, int32_t act,
                          float act_param, void* dz, double* dbias_sums, SpQ8 q8, sp_stream_t stream, int64_t y8_plane = 0) {
  SP_CHECK_ARG(y && (dz || q8.p) && CP % 8 == 0, "sp_pool_skip_act_bwd: bad arguments");
  SP_CHECK_ARG(y8_plane == 0 || (dtype == SP_BF16 && CP % 16 == 0 && y8_plane >= (int64_t)B * D * H * W * 16), "sp_pool_skip_act_bwd_y8: e4m3 planes of >= B D H W 16 bytes");
  SP_CHECK_ARG(!q8.p || (dtype == SP_BF16 && CP % 16 == 0 && q8.plane >= (int64_t)B * D * H * W * 16 && q8.scale > 0.f),
               "sp_pool_skip_act_bwd_q8: bf16 tensors of whole 16-channel planes");
  SP_CHECK_VOX((int64_t)B * D * H * W, "sp_pool_skip_act_bwd");
  SP_CHECK_ARG(!gp || coefp, "sp_pool_skip_act_bwd: pool gradient without coefficients");
  SP_CHECK_ARG(!gs || (coefs && cs0 % 8 == 0 && cs0 + CP <= CPcat && Dc <= D && Hc <= H && Wc <= W), "sp_pool_skip_act_bwd: bad skip arguments");
  if (coef_stride <= 0) { coef_stride = CPcat; coef_c0 = cs0; }      // coefficients laid out like the gradient tensor
  OctMap om = make_octmap(CP);
  Dims di{B, D, H, W}, dc{B, Dc, Hc, Wc};
  const int64_t nwin = (int64_t)B * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
  const unsigned grid = grid_for(nwin, om.vpb);
  const size_t sh = (size_t)CP * 7 * sizeof(float);
#define SP_L(A_)                                                                                                                          \
  if (dtype == SP_BF16 && y8_plane) hipLaunchKernelGGL((pool_skip_act_bwd_kernel<bf16_t, A_, true>), dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)y, \
                                           (const bf16_t*)gp, coefp, (const bf16_t*)cat, (const bf16_t*)gs, coefs, cs0, CPcat, coef_c0,   \
                                           coef_stride, di, CP, dc, om, act, act_param, (bf16_t*)dz, dbias_sums, q8, y8_plane);            \
  else if (dtype == SP_BF16) hipLaunchKernelGGL((pool_skip_act_bwd_kernel<bf16_t, A_>), dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)y, \
                                           (const bf16_t*)gp, coefp, (const bf16_t*)cat, (const bf16_t*)gs, coefs, cs0, CPcat, coef_c0,   \
                                           coef_stride, di, CP, dc, om, act, act_param, (bf16_t*)dz, dbias_sums, q8, y8_plane);            \
  else hipLaunchKernelGGL((pool_skip_act_bwd_kernel<float, A_>), dim3(grid), dim3(256), sh, ST(stream), (const float*)y, (const float*)gp, \
                          coefp, (const float*)cat, (const float*)gs, coefs, cs0, CPcat, coef_c0, coef_stride, di, CP, dc, om, act,       \
                          act_param, (float*)dz, dbias_sums, q8)
  SP_ACT_DISPATCH(act, SP_L)
#undef SP_L
  SP_CHECK_LAUNCH("sp_pool_skip_act_bwd");
  return SP_OK;
}
extern "C" int sp_pool_skip_act_bwd(const void* y, const void* gp, const float* coefp, const void* cat, const void* gs,
                                    const float* coefs, int32_t cs0, int32_t CPcat, int32_t coef_c0, int32_t coef_stride,
                                    int32_t dtype, int32_t B, int32_t D,
                                    int32_t H, int32_t W, int32_t CP, int32_t Dc, int32_t Hc, int32_t Wc, int32_t act,
                                    float act_param, void* dz, double* dbias_sums, sp_stream_t stream) {
  return pool_skip_impl(y, gp, coefp, cat, gs, coefs, cs0, CPcat, coef_c0, coef_stride, dtype, B, D, H, W, CP, Dc, Hc, Wc, act,
                        act_param, dz, dbias_sums, SpQ8{nullptr, 0, 1.f, 0}, stream);
}
// y given as its e4m3 plane-major copy (fp8 mode: the producing convolution stored no 16-bit tensor); q8 may be NULL
extern "C" int sp_pool_skip_act_bwd_y8(const void* y8, int64_t y8_plane, const void* gp, const float* coefp, const void* gs,
                                       const float* coefs, int32_t cs0, int32_t CPcat, int32_t coef_c0, int32_t coef_stride,
                                       int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, int32_t Dc, int32_t Hc, int32_t Wc,
                                       int32_t act, float act_param, void* dz, double* dbias_sums, void* q8, int64_t q8_plane,
                                       int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(y8_plane > 0 && (!q8 || q8_fmt == 0 || q8_fmt == 1), "sp_pool_skip_act_bwd_y8: bad arguments");
  return pool_skip_impl(y8, gp, coefp, nullptr, gs, coefs, cs0, CPcat, coef_c0, coef_stride, SP_BF16, B, D, H, W, CP, Dc, Hc, Wc, act,
                        act_param, dz, dbias_sums, SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8 ? q8_scale : 1.f, q8_fmt}, stream, y8_plane);
}
extern "C" int sp_pool_skip_act_bwd_q8(const void* y, const void* gp, const float* coefp, const void* cat, const void* gs,
                                       const float* coefs, int32_t cs0, int32_t CPcat, int32_t coef_c0, int32_t coef_stride,
                                       int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, int32_t Dc,
                                       int32_t Hc, int32_t Wc, int32_t act, float act_param, void* dz, double* dbias_sums,
                                       void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(q8 && (q8_fmt == 0 || q8_fmt == 1), "sp_pool_skip_act_bwd_q8: bad fp8 output");
  return pool_skip_impl(y, gp, coefp, cat, gs, coefs, cs0, CPcat, coef_c0, coef_stride, dtype, B, D, H, W, CP, Dc, Hc, Wc, act,
                        act_param, dz, dbias_sums, SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8_scale, q8_fmt}, stream);
}

// transposed trilinear x2: per axis, input i receives from outputs 2i-1 (w .25), 2i (.75, or 1 at i=0),
// 2i+1 (.75, or 1 at i=N-1), 2i+2 (.25)
__device__ __forceinline__ void upT_axis(int i, int N, int o[4], float w[4]) {
  o[0] = 2 * i - 1; w[0] = i >= 1 ? 0.25f : 0.f;
  o[1] = 2 * i;     w[1] = i >= 1 ? 0.75f : 1.f;
  o[2] = 2 * i + 1; w[2] = i < N - 1 ? 0.75f : 1.f;
  o[3] = 2 * i + 2; w[3] = i < N - 1 ? 0.25f : 0.f;
}
template <typename T>
__global__ __launch_bounds__(256) void upsample2_act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ cat,
                                                                 const T* __restrict__ g, const float* __restrict__ coef,
                                                                 int CPcat, int cstride, Dims di, int CP, OctMap om, int act, float ap,
                                                                 T* __restrict__ dz, double* __restrict__ dbias) {
  extern __shared__ float red[];
  const int slot = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - slot * om.OC;
  const bool active = slot < om.vpb;
  const int Do = 2 * di.D, Ho = 2 * di.H, Wo = 2 * di.W;
  const int64_t nin = (int64_t)di.B * di.D * di.H * di.W;
  float part[1][8], c0[8], c1[8], c2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    part[0][j] = 0.f;
    const int c = oc * 8 + j;
    c0[j] = active ? coef[c] : 0.f; c1[j] = active ? coef[cstride + c] : 0.f; c2[j] = active ? coef[2 * cstride + c] : 0.f;
  }
  const Unflat uf_(di.D, di.H, di.W);
  if (active) {
    // each workgroup walks ONE contiguous voxel range (neighbouring rows stay in its L1 / the XCD's L2)
    const int64_t chunk_ = ((nin + gridDim.x - 1) / gridDim.x + om.vpb - 1) / om.vpb * om.vpb;
    const int64_t vend_ = min((int64_t)nin, ((int64_t)blockIdx.x + 1) * chunk_);
    for (int64_t v = (int64_t)blockIdx.x * chunk_ + slot; v < vend_; v += om.vpb) {
      int b, z, yy, xx;
      uf_(v, b, z, yy, xx);
      int oz[4], oy[4], ox[4]; float wz[4], wy[4], wx[4];
      upT_axis(z, di.D, oz, wz); upT_axis(yy, di.H, oy, wy); upT_axis(xx, di.W, ox, wx);
      float d[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = 0.f;
      for (int a = 0; a < 4; ++a) {
        if (wz[a] == 0.f) continue;
        for (int bb = 0; bb < 4; ++bb) {
          if (wy[bb] == 0.f) continue;
          const float wzy = wz[a] * wy[bb];
          const int64_t rowbase = (((int64_t)b * Do + oz[a]) * Ho + oy[bb]) * Wo;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float w = wzy * wx[c];
            if (w != 0.f) {
              const int64_t o = (rowbase + ox[c]) * CPcat + oc * 8;
              float g8[8], c8[8];
              Store<T>::ld8(g + o, g8);
              Store<T>::ld8(cat + o, c8);
#pragma unroll
              for (int j = 0; j < 8; ++j) d[j] = fmaf(w, c0[j] * g8[j] + c1[j] * c8[j] + c2[j], d[j]);
            }
          }
        }
      }
      float y8[8];
      Store<T>::ld8(y + v * CP + oc * 8, y8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { d[j] *= act_bwd_from_y(act, ap, y8[j]); part[0][j] += d[j]; }
      Store<T>::st8(dz + v * CP + oc * 8, d);
    }
  }
  if (dbias) block_channel_reduce<1>(part, oc, active, CP, dbias, red);
}
// ---- tiled variant (channel octets dividing 256).  The gather above re-reads every cat-gradient voxel up to 8 times
// (4x4x4 overlapping windows; 1.6 GB fetched for 0.2 GB of data).  Here a workgroup owns a TY x TX patch of input
// columns and marches along z: each output plane pair is staged ONCE in LDS ((2TY+2) x (2TX+2) voxels, halo 1.2-1.4x),
// reduced in-plane (4x4 window) and folded into two running z accumulators.  The BatchNorm term c1 * sum_o w_o cat_o
// needs no second tensor: cat = U y, so sum_o w_o cat_o = (U^T U y)[v], a 3x3x3 stencil on y with per-axis weights
// (diag 1.25, 1.625 at the ends; off-diagonal 0.375), evaluated from a staged y patch; and sum_o w_o = 8.
struct UpTile {
  int TY, TX, RY, RX, nby, nbx, ZS, zchunk;
  FastDiv d_tx, d_rx, d_yx, d_nbx, d_nby, d_zs;
};
__device__ __forceinline__ float upM_diag(int i, int N) {
  int o[4]; float w[4];
  upT_axis(i, N, o, w);
  return w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3];
}
template <typename T, int ACT>
__global__ __launch_bounds__(256) void upsample2_act_bwd_tiled_kernel(const T* __restrict__ y, const T* __restrict__ g,
                                                                       const float* __restrict__ coef, int CPcat, int cstride, Dims di,
                                                                       int CP, OctMap om, UpTile ut, int act, float ap,
                                                                       T* __restrict__ dz, double* __restrict__ dbias, const SpQ8 q8) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int OC = om.OC, TY = ut.TY, TX = ut.TX, RY = ut.RY, RX = ut.RX;
  const int pos = fdiv(threadIdx.x, om.d_oc), oc = threadIdx.x - pos * OC;
  const int ty = fdiv(pos, ut.d_tx), tx = pos - ty * TX;
  // block -> (b, by, bx, zc)
  uint32_t bid = blockIdx.x;
  const uint32_t q1 = fdiv(bid, ut.d_zs); const int zc = bid - q1 * ut.ZS;
  const uint32_t q2 = fdiv(q1, ut.d_nbx); const int bx = q1 - q2 * ut.nbx;
  const uint32_t q3 = fdiv(q2, ut.d_nby); const int by = q2 - q3 * ut.nby;
  const int b = q3;
  const int D = di.D, H = di.H, W = di.W, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const int z0 = zc * ut.zchunk, z1 = min(D, z0 + ut.zchunk);
  const int y0 = by * TY, x0 = bx * TX;
  const int yy = y0 + ty, xx = x0 + tx;
  const bool valid = yy < H && xx < W;
  const int yc_ = min(yy, H - 1), xc_ = min(xx, W - 1);      // clamped twin for the weights of idle threads

  const size_t gplane = (size_t)RY * RX * CP * sizeof(T);     // one staged cat-gradient plane
  T* gb0 = reinterpret_cast<T*>(smem);
  T* gb1 = reinterpret_cast<T*>(smem + gplane);
  T* yb = reinterpret_cast<T*>(smem + 2 * gplane);
  float* red = reinterpret_cast<float*>(smem + 2 * gplane + (size_t)(TY + 2) * (TX + 2) * CP * sizeof(T));

  int oy[4], ox[4]; float wy[4], wx[4];
  upT_axis(yc_, H, oy, wy); upT_axis(xc_, W, ox, wx);
  const float myc = upM_diag(yc_, H), mxc = upM_diag(xc_, W);
  const float my[3] = {yc_ >= 1 ? 0.375f : 0.f, myc, yc_ < H - 1 ? 0.375f : 0.f};
  const float mx[3] = {xc_ >= 1 ? 0.375f : 0.f, mxc, xc_ < W - 1 ? 0.375f : 0.f};
  float c0[8], c1[8], c2[8], part[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = oc * 8 + j;
    c0[j] = coef[c]; c1[j] = coef[cstride + c]; c2[j] = 8.f * coef[2 * cstride + c];
    part[0][j] = 0.f;
  }
  float Acur[8], Anext[8], qprev[8], qcur[8], qnext[8], ycur[8], ynext[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) Acur[j] = Anext[j] = qprev[j] = qcur[j] = qnext[j] = ycur[j] = ynext[j] = 0.f;

  const int nreg = RY * RX, nyp = (TY + 2) * (TX + 2), vstep = 256 / OC;
  for (int m = z0 - 2; m < z1; ++m) {
    const bool yplane = m + 1 >= 0 && m + 1 < D;
    {
      // ---------------- stage: cat-gradient planes 2m+1, 2m+2 (clamped coordinates carry zero weight) and y plane m+1
      if (m >= z0 - 1) {
        const int oza = min(max(2 * m + 1, 0), Do - 1), ozb = min(max(2 * m + 2, 0), Do - 1);
        for (int vi = pos; vi < nreg; vi += vstep) {
          const int vy = fdiv(vi, ut.d_rx), vx = vi - vy * RX;
          const int gy = min(max(2 * y0 - 1 + vy, 0), Ho - 1), gx = min(max(2 * x0 - 1 + vx, 0), Wo - 1);
          const size_t o_a = ((((size_t)b * Do + oza) * Ho + gy) * Wo + gx) * CPcat + oc * 8;
          const size_t o_b = ((((size_t)b * Do + ozb) * Ho + gy) * Wo + gx) * CPcat + oc * 8;
          const int lo = vi * CP + oc * 8;
          if constexpr (sizeof(T) == 2) {
            const uint4 va = *reinterpret_cast<const uint4*>(g + o_a), vb = *reinterpret_cast<const uint4*>(g + o_b);
            *reinterpret_cast<uint4*>(gb0 + lo) = va; *reinterpret_cast<uint4*>(gb1 + lo) = vb;
          } else {
            const uint4 va0 = *reinterpret_cast<const uint4*>(g + o_a), va1 = *reinterpret_cast<const uint4*>(g + o_a + 4);
            const uint4 vb0 = *reinterpret_cast<const uint4*>(g + o_b), vb1 = *reinterpret_cast<const uint4*>(g + o_b + 4);
            *reinterpret_cast<uint4*>(gb0 + lo) = va0; *reinterpret_cast<uint4*>(gb0 + lo + 4) = va1;
            *reinterpret_cast<uint4*>(gb1 + lo) = vb0; *reinterpret_cast<uint4*>(gb1 + lo + 4) = vb1;
          }
        }
      }
      if (yplane) {
        for (int vi = pos; vi < nyp; vi += vstep) {
          const int vy = fdiv(vi, ut.d_yx), vx = vi - vy * (TX + 2);
          const int sy = min(max(y0 - 1 + vy, 0), H - 1), sx = min(max(x0 - 1 + vx, 0), W - 1);
          const size_t o = ((((size_t)b * D + (m + 1)) * H + sy) * W + sx) * CP + oc * 8;
          const int lo = vi * CP + oc * 8;
          if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(yb + lo) = *reinterpret_cast<const uint4*>(y + o);
          } else {
            *reinterpret_cast<uint4*>(yb + lo) = *reinterpret_cast<const uint4*>(y + o);
            *reinterpret_cast<uint4*>(yb + lo + 4) = *reinterpret_cast<const uint4*>(y + o + 4);
          }
        }
      }
      __syncthreads();
    }
    // ---------------- in-plane 4x4 reductions
    if (m >= z0 - 1) {
      float P1[8], P2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) P1[j] = P2[j] = 0.f;
      // rows rolled on purpose (weight picked by selects): fully unrolled, hipcc hoists all 32 LDS reads and the kernel
      // needs 294 VGPRs -- one workgroup per CU, every plane pair then waits out its own HBM latency
#pragma unroll 1
      for (int bb = 0; bb < 4; ++bb) {
        const float wyb = bb == 0 ? wy[0] : (bb == 1 ? wy[1] : (bb == 2 ? wy[2] : wy[3]));
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float w = wyb * wx[cc];
          const int lo = ((2 * ty + bb) * RX + 2 * tx + cc) * CP + oc * 8;
          float a8[8], b8[8];
          Store<T>::ld8(gb0 + lo, a8);
          Store<T>::ld8(gb1 + lo, b8);
#pragma unroll
          for (int j = 0; j < 8; ++j) { P1[j] = fmaf(w, a8[j], P1[j]); P2[j] = fmaf(w, b8[j], P2[j]); }
        }
      }
      int oz[4]; float wz[4], wn[4];
      upT_axis(max(m, 0), D, oz, wz);
      upT_axis(min(m + 1, D - 1), D, oz, wn);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Acur[j] += wz[2] * P1[j] + wz[3] * P2[j];
        Anext[j] = wn[0] * P1[j] + wn[1] * P2[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) qnext[j] = ynext[j] = 0.f;
    if (yplane) {
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy) {
        const float myd = dy == 0 ? my[0] : (dy == 1 ? my[1] : my[2]);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float w = myd * mx[dx];
          float v8[8];
          Store<T>::ld8(yb + ((ty + dy) * (TX + 2) + tx + dx) * CP + oc * 8, v8);
#pragma unroll
          for (int j = 0; j < 8; ++j) qnext[j] = fmaf(w, v8[j], qnext[j]);
          if (dy == 1 && dx == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ynext[j] = v8[j];
          }
        }
      }
    }
    // ---------------- input plane m is complete
    if (m >= z0) {
      const float mzc = upM_diag(m, D);
      const float mzl = m >= 1 ? 0.375f : 0.f, mzr = m < D - 1 ? 0.375f : 0.f;
      float o8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float my_ = mzl * qprev[j] + mzc * qcur[j] + mzr * qnext[j];
        o8[j] = (c0[j] * Acur[j] + c1[j] * my_ + c2[j]) * act_bwd_t<ACT>(act, ap, ycur[j]);
      }
      if (valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[0][j] += o8[j];
        const size_t vo_ = (((size_t)b * D + m) * H + yy) * W + xx;
        if (dz) Store<T>::st8(dz + vo_ * CP + oc * 8, o8);      // (NULL: both readers take the fp8 copy)
        if (q8.p) sp_q8_store8(q8, (int64_t)vo_, oc, o8);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      Acur[j] = Anext[j]; qprev[j] = qcur[j]; qcur[j] = qnext[j]; ycur[j] = ynext[j];
    }
    __syncthreads();
  }
  if (dbias) block_channel_reduce<1>(part, oc, true, CP, dbias, red);
}

// ---- ring variant of the tiled kernel (bf16, 16 / 32 / 64 channels).  The tiled kernel above stages a plane pair with
// ordinary loads and waits for them at a barrier before every reduction: 1.8 TB/s (135 us for the 32-channel part of the
// last concat gradient, 75 us for the 64-channel one).  Here the planes stream through a RING of four LDS slots filled by
// LDS-DMA three planes ahead (counted s_waitcnt, one barrier per plane, border chunks by clamped addresses exactly as
// above), the y patch through two slots of its own, and the work -- (column, input plane) pairs -- is cut into gridDim.x
// equal pieces like the weight-gradient kernels do, so one workgroup per CU (the ring fills most of the LDS) still ends
// the launch together.  Arithmetic and summation order are those of the tiled kernel.
template <int OCT, int ACT>
__global__ __launch_bounds__(256) void upsample2_act_bwd_ring_kernel(const bf16_t* __restrict__ y, const bf16_t* __restrict__ g,
                                                                      const float* __restrict__ coef, int CPcat, int cstride,
                                                                      Dims di, int nby, int nbx, int act, float ap,
                                                                      bf16_t* __restrict__ dz, double* __restrict__ dbias, const SpQ8 q8,
                                                                      int CPy) {
  // CPy: channel pitch of y / dz (and width of the coefficient rows); blockIdx.y = group of CP channels of it (128 / 256-channel
  // tensors run as 2 / 4 groups of 64: the tile of a wider group would not fit the ring)
  constexpr int CP = OCT * 8, TX = 16, TY = 256 / (16 * OCT), RY = 2 * TY + 2, RX = 2 * TX + 2;
  const int cg0 = blockIdx.y * CP;
  constexpr int NCH = RY * RX * OCT, NJ = (NCH + 255) / 256, PSB = NJ * 4096;          // plane slot
  constexpr int NYC = (TY + 2) * (TX + 2) * OCT, NJY = (NYC + 255) / 256, YSB = NJY * 4096;
  constexpr int NS = 4, DP = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* pring = smem;
  unsigned char* yring = smem + NS * PSB;
  float* red = nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pos = tid / OCT, oc = tid - pos * OCT;
  const int ty = pos / TX, tx = pos - ty * TX;
  const int D = di.D, H = di.H, W = di.W, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  float c0[8], c1[8], c2[8], part[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg0 + oc * 8 + j;
    c0[j] = coef[c]; c1[j] = coef[cstride + c]; c2[j] = 8.f * coef[2 * cstride + c];
    part[0][j] = 0.f;
  }
  const int64_t gplane_b = (int64_t)Ho * Wo * CPcat * 2, yplane_b = (int64_t)H * W * CPy * 2;
  const uint32_t ncols = (uint32_t)di.B * nby * nbx;
  const uint64_t T = (uint64_t)ncols * D;
  uint64_t wpos = T * blockIdx.x / gridDim.x;
  const uint64_t wend = T * (blockIdx.x + 1) / gridDim.x;
  while (wpos < wend) {
    const uint32_t col = (uint32_t)(wpos / (uint32_t)D);
    const int z0 = (int)(wpos - (uint64_t)col * D);
    const int z1 = (int)min((uint64_t)D, (uint64_t)z0 + (wend - wpos));
    wpos += (uint64_t)(z1 - z0);
    const int bx = col % nbx, by = (col / nbx) % nby, b = col / (nbx * nby);
    const int y0 = by * TY, x0 = bx * TX;
    const int yy = y0 + ty, xx = x0 + tx;
    const bool valid = yy < H && xx < W;
    const int yc_ = min(yy, H - 1), xc_ = min(xx, W - 1);
    int oy[4], ox[4]; float wy[4], wx[4];
    upT_axis(yc_, H, oy, wy); upT_axis(xc_, W, ox, wx);
    const float myc = upM_diag(yc_, H), mxc = upM_diag(xc_, W);
    const float my[3] = {yc_ >= 1 ? 0.375f : 0.f, myc, yc_ < H - 1 ? 0.375f : 0.f};
    const float mx[3] = {xc_ >= 1 ? 0.375f : 0.f, mxc, xc_ < W - 1 ? 0.375f : 0.f};
    // per-lane DMA plans of this column: byte offsets inside a plane (clamped coordinates carry zero weight)
    uint32_t relp[NJ], rely[NJY];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int c = (wave + 4 * j) * 64 + lane;
      c = c < NCH ? c : NCH - 1;
      const int vi = c / OCT, o_ = c - vi * OCT, vy = vi / RX, vx = vi - vy * RX;
      const int gy = min(max(2 * y0 - 1 + vy, 0), Ho - 1), gx = min(max(2 * x0 - 1 + vx, 0), Wo - 1);
      relp[j] = (uint32_t)(((gy * Wo + gx) * CPcat + cg0 + o_ * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < NJY; ++j) {
      int c = (wave + 4 * j) * 64 + lane;
      c = c < NYC ? c : NYC - 1;
      const int vi = c / OCT, o_ = c - vi * OCT, vy = vi / (TX + 2), vx = vi - vy * (TX + 2);
      const int sy = min(max(y0 - 1 + vy, 0), H - 1), sx = min(max(x0 - 1 + vx, 0), W - 1);
      rely[j] = (uint32_t)(((sy * W + sx) * CPy + cg0 + o_ * 8) * 2);
    }
    const unsigned char* gb_ = reinterpret_cast<const unsigned char*>(g) + (int64_t)b * Do * gplane_b;
    const unsigned char* yb_ = reinterpret_cast<const unsigned char*>(y) + (int64_t)b * D * yplane_b;
    // steps s = 0 .. S-1 handle m = z0 - 2 + s; planes j = 0 .. 2(S-1)-1 belong to step 1 + j/2 (output planes 2m+1, 2m+2);
    // the y plane of step s (input plane m + 1) travels with plane j = 2s - 1, the one of step 0 alone in the prologue
    const int S = z1 - z0 + 2;
    auto issue_plane = [&](int j) {                       // plane j of the sequence into slot j & 3 (+ a y plane when j is odd)
      const int m = z0 - 1 + (j >> 1);
      const int zo = min(max(2 * m + 1 + (j & 1), 0), Do - 1);
      const unsigned char* src = gb_ + (int64_t)zo * gplane_b;
      unsigned char* dst = pring + (j & (NS - 1)) * PSB + wave * 1024;
#pragma unroll
      for (int k = 0; k < NJ; ++k) sp_dma16_nc(src + relp[k], dst + k * 4096);
      if (j & 1) {
        const int s = (j + 1) >> 1;
        const int zy = min(max(z0 - 2 + s + 1, 0), D - 1);
        const unsigned char* ys = yb_ + (int64_t)zy * yplane_b;
        unsigned char* yd = yring + (s & 1) * YSB + wave * 1024;
#pragma unroll
        for (int k = 0; k < NJY; ++k) sp_dma16_nc(ys + rely[k], yd + k * 4096);
      }
    };
    float Acur[8], Anext[8], qprev[8], qcur[8], qnext[8], ycur[8], ynext[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) Acur[j] = Anext[j] = qprev[j] = qcur[j] = qnext[j] = ycur[j] = ynext[j] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // the previous piece's slots are consumed
    {
      const int zy = min(max(z0 - 1, 0), D - 1);            // y plane of step 0
      const unsigned char* ys = yb_ + (int64_t)zy * yplane_b;
#pragma unroll
      for (int k = 0; k < NJY; ++k) sp_dma16_nc(ys + rely[k], yring + wave * 1024 + k * 4096);
    }
    issue_plane(0); issue_plane(1); issue_plane(2);
    for (int s = 0; s < S; ++s) {
      const int m = z0 - 2 + s;
      const bool yplane = m + 1 >= 0 && m + 1 < D;
      float P1[8], P2[8];
      f2_t Pp1[4], Pp2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) Pp1[j] = Pp2[j] = f2_splat(0.f);
      if (s == 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NJ + NJY) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * (s - 1) + h;
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ + NJY) : "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          issue_plane(j + DP);
          const bf16_t* gb = reinterpret_cast<const bf16_t*>(pring + (j & (NS - 1)) * PSB);
#pragma unroll 1
          for (int bb = 0; bb < 4; ++bb) {
            const float wyb = bb == 0 ? wy[0] : (bb == 1 ? wy[1] : (bb == 2 ? wy[2] : wy[3]));
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const f2_t w2 = f2_splat(wyb * wx[cc]);
              RawOct<bf16_t> ro;
              ro.load(gb + ((2 * ty + bb) * RX + 2 * tx + cc) * CP + oc * 8);
              f2_t a2[4];
              raw_get2(ro, a2);                      // packed pairs: v_pk_fma_f32 halves the multiply-adds of this VALU-bound loop
              if (h == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Pp1[q] = f2_fma(w2, a2[q], Pp1[q]);
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) Pp2[q] = f2_fma(w2, a2[q], Pp2[q]);
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { P1[2 * j] = Pp1[j].x; P1[2 * j + 1] = Pp1[j].y; P2[2 * j] = Pp2[j].x; P2[2 * j + 1] = Pp2[j].y; }
        int oz[4]; float wz[4], wn[4];
        upT_axis(max(m, 0), D, oz, wz);
        upT_axis(min(m + 1, D - 1), D, oz, wn);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          Acur[j] += wz[2] * P1[j] + wz[3] * P2[j];
          Anext[j] = wn[0] * P1[j] + wn[1] * P2[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) qnext[j] = ynext[j] = 0.f;
      if (yplane) {
        const bf16_t* yb = reinterpret_cast<const bf16_t*>(yring + (s & 1) * YSB);
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
          const float myd = dy == 0 ? my[0] : (dy == 1 ? my[1] : my[2]);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float w = myd * mx[dx];
            float v8[8];
            Store<bf16_t>::ld8(yb + ((ty + dy) * (TX + 2) + tx + dx) * CP + oc * 8, v8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qnext[j] = fmaf(w, v8[j], qnext[j]);
            if (dy == 1 && dx == 1) {
#pragma unroll
              for (int j = 0; j < 8; ++j) ynext[j] = v8[j];
            }
          }
        }
      }
      if (m >= z0) {
        const float mzc = upM_diag(m, D);
        const float mzl = m >= 1 ? 0.375f : 0.f, mzr = m < D - 1 ? 0.375f : 0.f;
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float my_ = mzl * qprev[j] + mzc * qcur[j] + mzr * qnext[j];
          o8[j] = (c0[j] * Acur[j] + c1[j] * my_ + c2[j]) * act_bwd_t<ACT>(act, ap, ycur[j]);
        }
        if (valid) {
#pragma unroll
          for (int j = 0; j < 8; ++j) part[0][j] += o8[j];
          const size_t vo_ = (((size_t)b * D + m) * H + yy) * W + xx;
          if (dz) Store<bf16_t>::st8(dz + vo_ * CPy + cg0 + oc * 8, o8);
          if (q8.p) sp_q8_store8(q8, (int64_t)vo_, blockIdx.y * OCT + oc, o8);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Acur[j] = Anext[j]; qprev[j] = qcur[j]; qcur[j] = qnext[j]; ycur[j] = ynext[j];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (dbias) block_channel_reduce<1>(part, oc, true, CP, dbias + cg0, red, CPy);
}

template <int OCT>
static int launch_up_bwd_ring(const void* y, const void* g, const float* coef, int CPcat, int cstride, Dims di, int act, float ap,
                              void* dz, double* dbias, SpQ8 q8, hipStream_t st, int groups = 1) {
  constexpr int TY = 256 / (16 * OCT), NJ = ((2 * TY + 2) * 34 * OCT + 255) / 256, NJY = ((TY + 2) * 18 * OCT + 255) / 256;
  const int lds = 4 * NJ * 4096 + 2 * NJY * 4096;
  const int nby = (di.H + TY - 1) / TY, nbx = (di.W + 15) / 16;
  const int64_t T = (int64_t)di.B * nby * nbx * di.D;
  const int per = 256 / groups;                       // one workgroup per CU over all channel groups
  const dim3 grid((unsigned)(T < per ? T : per), (unsigned)groups);
  const int CPy = OCT * 8 * groups;
#define SP_L(A_)                                                                                                       \
  {                                                                                                                    \
    auto kern = upsample2_act_bwd_ring_kernel<OCT, A_>;                                                                \
    SP_ENSURE_LDS(kern, lds, "sp_upsample2_act_bwd");                                                                  \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const bf16_t*)y, (const bf16_t*)g, coef, CPcat, cstride, di, nby, nbx, \
                       act, ap, (bf16_t*)dz, dbias, q8, CPy);                                                          \
  }
  SP_ACT_DISPATCH(act, SP_L)
#undef SP_L
  SP_CHECK_LAUNCH("sp_upsample2_act_bwd(ring)");
  return SP_OK;
}

static int upsample2_act_bwd_impl(const void* y, const void* cat, const void* g, const float* coef, int32_t CPcat,
                                  int32_t coef_stride, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, int32_t act,
                                  float act_param, void* dz, double* dbias_sums, SpQ8 q8, sp_stream_t stream) {
  SP_CHECK_ARG(y && g && coef && (dz || q8.p) && CP % 8 == 0 && CPcat >= CP, "sp_upsample2_act_bwd: bad arguments");
  SP_CHECK_ARG(!q8.p || (dtype == SP_BF16 && CP % 16 == 0 && q8.scale > 0.f && q8.plane >= (int64_t)B * D * H * W * 16),
               "sp_upsample2_act_bwd_q8: bf16 tensors of whole 16-channel planes");
  if (coef_stride <= 0) coef_stride = CPcat;
  SP_CHECK_VOX((int64_t)B * D * H * W * 8, "sp_upsample2_act_bwd");
  OctMap om = make_octmap(CP);
  Dims di{B, D, H, W};
  static const bool ring_groups_ = !getenv("SP_UPSAMPLE_BWD_NO_GROUPS");      // (A/B knob: 128 / 256 channels on the tiled kernel)
  if (dtype == SP_BF16 && (CP == 16 || CP == 32 || CP == 64 || (ring_groups_ && (CP == 128 || CP == 256))) && D >= 2 && H >= 2 && W >= 2 &&
      (int64_t)4 * H * W * CPcat * 2 < (1ll << 31) && !getenv("SP_UPSAMPLE_BWD_TILED") && !getenv("SP_UPSAMPLE_BWD_GATHER")) {
    if (CP == 16) return launch_up_bwd_ring<2>(y, g, coef, CPcat, coef_stride, di, act, act_param, dz, dbias_sums, q8, ST(stream));
    if (CP == 32) return launch_up_bwd_ring<4>(y, g, coef, CPcat, coef_stride, di, act, act_param, dz, dbias_sums, q8, ST(stream));
    // 64 channels as two groups of 32: the 4-row tile of the 32-channel instance has less halo (1.33x against 1.59x), and the
    // two groups of a position run side by side on one XCD (workgroup ids 128 apart), sharing the lines they both touch
    // (64 -> 32 @84^3 + 168^3: 596 -> 559 us; fp8 4-scale step 19.8 -> 19.6 ms)
    static const bool split64_ = getenv("SP_UPBWD_NO_SPLIT64") == nullptr;
    if (split64_ && CP == 64) return launch_up_bwd_ring<4>(y, g, coef, CPcat, coef_stride, di, act, act_param, dz, dbias_sums, q8, ST(stream), 2);
    return launch_up_bwd_ring<8>(y, g, coef, CPcat, coef_stride, di, act, act_param, dz, dbias_sums, q8, ST(stream), CP / 64);
  }
  if (256 % om.OC == 0 && D >= 2 && H >= 2 && W >= 2 && !getenv("SP_UPSAMPLE_BWD_GATHER")) {
    // tiled path: TY x TX input columns per workgroup, z split into chunks so that ~3 workgroups per CU exist
    UpTile ut;
    const int npos = 256 / om.OC;
    ut.TX = npos < 16 ? npos : 16; ut.TY = npos / ut.TX;
    ut.RY = 2 * ut.TY + 2; ut.RX = 2 * ut.TX + 2;
    ut.nby = (H + ut.TY - 1) / ut.TY; ut.nbx = (W + ut.TX - 1) / ut.TX;
    const int cols = B * ut.nby * ut.nbx;
    int ZS = (768 + cols - 1) / cols;
    if (ZS > D / 4) ZS = D / 4;
    if (ZS < 1) ZS = 1;
    ut.zchunk = (D + ZS - 1) / ZS; ut.ZS = (D + ut.zchunk - 1) / ut.zchunk;
    ut.d_tx = make_fastdiv(ut.TX); ut.d_rx = make_fastdiv(ut.RX); ut.d_yx = make_fastdiv(ut.TX + 2);
    ut.d_nbx = make_fastdiv(ut.nbx); ut.d_nby = make_fastdiv(ut.nby); ut.d_zs = make_fastdiv(ut.ZS);
    const size_t esz = dtype == SP_BF16 ? 2 : 4;
    const size_t sh2 = (2 * (size_t)ut.RY * ut.RX + (size_t)(ut.TY + 2) * (ut.TX + 2)) * CP * esz + (size_t)CP * sizeof(float);
    if (sh2 <= 150 * 1024) {      // + 8 KB static (block_channel_reduce)
      const unsigned grid2 = (unsigned)(cols * ut.ZS);
#define SP_L(A_)                                                                                                                      \
  if (dtype == SP_BF16) {                                                                                                             \
    auto kern = upsample2_act_bwd_tiled_kernel<bf16_t, A_>;                                                                           \
    SP_ENSURE_LDS(kern, (int)sh2, "sp_upsample2_act_bwd");                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid2), dim3(256), sh2, ST(stream), (const bf16_t*)y, (const bf16_t*)g, coef, CPcat, coef_stride, di, \
                       CP, om, ut, act, act_param, (bf16_t*)dz, dbias_sums, q8);                                                     \
  } else {                                                                                                                            \
    auto kern = upsample2_act_bwd_tiled_kernel<float, A_>;                                                                            \
    SP_ENSURE_LDS(kern, (int)sh2, "sp_upsample2_act_bwd");                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid2), dim3(256), sh2, ST(stream), (const float*)y, (const float*)g, coef, CPcat, coef_stride, di,  \
                       CP, om, ut, act, act_param, (float*)dz, dbias_sums, q8);                                                      \
  }
      SP_ACT_DISPATCH(act, SP_L)
#undef SP_L
      SP_CHECK_LAUNCH("sp_upsample2_act_bwd");
      return SP_OK;
    }
  }
  SP_CHECK_ARG(cat && dz && !q8.p, "sp_upsample2_act_bwd: the gather fallback needs the concat buffer (same pitch as g) and writes no fp8 copy");
  const unsigned grid = grid_for((int64_t)B * D * H * W, om.vpb);
  const size_t sh = (size_t)CP * sizeof(float);
  if (dtype == SP_BF16) hipLaunchKernelGGL(upsample2_act_bwd_kernel<bf16_t>, dim3(grid), dim3(256), sh, ST(stream), (const bf16_t*)y, (const bf16_t*)cat, (const bf16_t*)g, coef, CPcat, coef_stride, di, CP, om, act, act_param, (bf16_t*)dz, dbias_sums);
  else hipLaunchKernelGGL(upsample2_act_bwd_kernel<float>, dim3(grid), dim3(256), sh, ST(stream), (const float*)y, (const float*)cat, (const float*)g, coef, CPcat, coef_stride, di, CP, om, act, act_param, (float*)dz, dbias_sums);
  SP_CHECK_LAUNCH("sp_upsample2_act_bwd");
  return SP_OK;
}
extern "C" int sp_upsample2_act_bwd(const void* y, const void* cat, const void* g, const float* coef, int32_t CPcat,
                                    int32_t coef_stride, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, int32_t act,
                                    float act_param, void* dz, double* dbias_sums, sp_stream_t stream) {
  return upsample2_act_bwd_impl(y, cat, g, coef, CPcat, coef_stride, dtype, B, D, H, W, CP, act, act_param, dz, dbias_sums,
                                SpQ8{nullptr, 0, 1.f, 0}, stream);
}
extern "C" int sp_upsample2_act_bwd_q8(const void* y, const void* cat, const void* g, const float* coef, int32_t CPcat,
                                       int32_t coef_stride, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP,
                                       int32_t act, float act_param, void* dz /* or NULL */, double* dbias_sums, void* q8,
                                       int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream) {
  SP_CHECK_ARG(q8 && (q8_fmt == 0 || q8_fmt == 1), "sp_upsample2_act_bwd_q8: bad fp8 output");
  return upsample2_act_bwd_impl(y, cat, g, coef, CPcat, coef_stride, dtype, B, D, H, W, CP, act, act_param, dz, dbias_sums,
                                SpQ8{reinterpret_cast<unsigned char*>(q8), q8_plane, q8_scale, q8_fmt}, stream);
}

// ------------------------------------------------------------------------------------------------ network output side
// dz[b,v,c] = dout[b,c,v] * act'(out[b,c,v]) ; NCDHW fp32 -> channels-last ; dbias_sums[c] += sum dz
template <typename T>
__global__ __launch_bounds__(256) void out_grad_to_cl_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                              int C, int64_t DHW, int CP, int64_t total, int act, float ap,
                                                              T* __restrict__ dz, double* __restrict__ dbias) {
  __shared__ float red[4 * 8];      // [wave][channel]
  float part[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[j] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / DHW, v = i - b * DHW;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = 0.f;
      if (j < C) {
        const int64_t o = (b * C + j) * DHW + v;
        f[j] = dout[o] * act_bwd_from_y(act, ap, out[o]);
        part[j] += f[j];
      }
    }
    Store<T>::st8(dz + i * CP, f);
    for (int c0 = 8; c0 < CP; c0 += 8) {
      float zz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      Store<T>::st8(dz + i * CP + c0, zz);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s = wave_sum(part[j]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 8 + j] = s;      // (ordered: sp_cols_sum)
  }
  __syncthreads();
  if (dbias && threadIdx.x < C) atomicAdd(&dbias[(size_t)(blockIdx.x % SP_REDUCE_ROWS) * CP + threadIdx.x], (double)sp_cols_sum(red, 8, 4, threadIdx.x));
}
extern "C" int sp_out_grad_to_cl(const float* dout, const float* out, int32_t B, int32_t C, int64_t DHW, int32_t CP,
                                 int32_t dtype, int32_t act, float act_param, void* dz, double* dbias_sums,
                                 sp_stream_t stream) {
  SP_CHECK_ARG(dout && out && dz && C >= 1 && C <= 8 && CP % 8 == 0, "sp_out_grad_to_cl: needs 1..8 output channels");
  const int64_t total = (int64_t)B * DHW;
  const unsigned grid = (unsigned)((total + 255) / 256 > MAX_BLOCKS ? MAX_BLOCKS : (total + 255) / 256);
  if (dtype == SP_BF16) hipLaunchKernelGGL(out_grad_to_cl_kernel<bf16_t>, dim3(grid), dim3(256), 0, ST(stream), dout, out, C, DHW, CP, total, act, act_param, (bf16_t*)dz, dbias_sums);
  else hipLaunchKernelGGL(out_grad_to_cl_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), dout, out, C, DHW, CP, total, act, act_param, (float*)dz, dbias_sums);
  SP_CHECK_LAUNCH("sp_out_grad_to_cl");
  return SP_OK;
}

// BatchDiceLoss pieces (metrics.py:16-28): sums[c] = (sum o*t, sum o*o, sum t*t) over batch and volume.
// o / t are (B, C, DHW) with an arbitrary BATCH stride (elements): dto.outputs.core / .penu are channel slices of one
// (B, 2, DHW) tensor and are read in place.
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ o, int64_t obs, const float* __restrict__ t,
                                                         int64_t tbs, int C, int64_t DHW, double* __restrict__ sums) {
  // grid.y = b*C + c ; grid.x strides over the volume
  const int bc = blockIdx.y, c = bc % C, b = bc / C;
  const float* op = o + (int64_t)b * obs + (int64_t)c * DHW;
  const float* tp = t + (int64_t)b * tbs + (int64_t)c * DHW;
  float s[3] = {0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < DHW; i += (int64_t)gridDim.x * 256) {
    const float a = op[i], bb = tp[i];
    s[0] += a * bb; s[1] += a * a; s[2] += bb * bb;
  }
  __shared__ float red[4 * 3];      // [wave][moment], added up in wave order (sp_cols_sum)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float w = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 3 + k] = w;
  }
  __syncthreads();
  // replica row per workgroup (rows 128 bytes or more apart): 1024 same-line fp64 atomics cost ~20 us at the tail
  if (threadIdx.x < 3)
    atomicAdd(&sums[(size_t)((blockIdx.x + blockIdx.y) % SP_REDUCE_ROWS) * SP_DICE_PITCH(C) + c * 3 + threadIdx.x], (double)sp_cols_sum(red, 3, 4, threadIdx.x));
}
extern "C" int sp_dice_sums(const float* o, int64_t o_bstride, const float* t, int64_t t_bstride, int32_t B, int32_t C,
                            int64_t DHW, double* sums, sp_stream_t stream) {
  SP_CHECK_ARG(o && t && sums && B >= 1 && C >= 1 && o_bstride >= C * DHW && t_bstride >= C * DHW, "sp_dice_sums: bad arguments");
  int64_t gx = (DHW + 256 * 8 - 1) / (256 * 8);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(dice_sums_kernel, dim3((unsigned)gx, B * C), dim3(256), 0, ST(stream), o, o_bstride, t, t_bstride, C, DHW, sums);
  SP_CHECK_LAUNCH("sp_dice_sums");
  return SP_OK;
}
// loss = 1 - sum_c w_c (2 I_c + eps) / (O_c + T_c + eps);  coef[c] = (ca, cb) with d loss / d o = ca*t + cb*o:
// ca = -2 w / den, cb = 2 w num / den^2.  One launch instead of a dozen one-element torch kernels.
// clear != NULL (= sums): the replica rows are zeroed again once they are read -- the caller keeps ONE accumulator and needs no fill
// launch in front of the next sp_dice_sums (4.9 us of a training step's dependent chain)
__global__ void dice_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ w, double eps, int C,
                                     float* __restrict__ loss, float* __restrict__ coef, double* __restrict__ clear) {
  const int pitch = SP_DICE_PITCH(C);
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int c = 0; c < C; ++c) {
      const double num = 2.0 * sp_rows_sum(sums, c * 3, pitch) + eps;
      const double den = sp_rows_sum(sums, c * 3 + 1, pitch) + sp_rows_sum(sums, c * 3 + 2, pitch) + eps;
      acc += (double)w[c] * num / den;
      coef[2 * c] = (float)(-2.0 * w[c] / den);
      coef[2 * c + 1] = (float)(2.0 * w[c] * num / (den * den));
    }
    *loss = (float)(1.0 - acc);
  }
  if (clear) {
    __syncthreads();
    for (int k = threadIdx.x; k < SP_REDUCE_ROWS * pitch; k += blockDim.x) clear[k] = 0.0;
  }
}
extern "C" int sp_dice_finalize(const double* sums, const float* weights, double eps, int32_t C, float* loss, float* coef,
                                sp_stream_t stream) {
  SP_CHECK_ARG(sums && weights && loss && coef && C >= 1, "sp_dice_finalize: bad arguments");
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), sums, weights, eps, C, loss, coef, (double*)nullptr);
  SP_CHECK_LAUNCH("sp_dice_finalize");
  return SP_OK;
}
extern "C" int sp_dice_finalize_clear(double* sums, const float* weights, double eps, int32_t C, float* loss, float* coef,
                                      sp_stream_t stream) {
  SP_CHECK_ARG(sums && weights && loss && coef && C >= 1, "sp_dice_finalize_clear: bad arguments");
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), sums, weights, eps, C, loss, coef, sums);
  SP_CHECK_LAUNCH("sp_dice_finalize_clear");
  return SP_OK;
}

// ---- the CAE reconstruction loss as three launches (CaeReconstructionLearner.py:52-70):
//   [ mean(|p - i| - (p - i)) + mean(|p - c| - (p - c)) + Dice(c, tc) + Dice(p, tp) + Dice(l, tl) + f mean|zi - zl| ] / (5 + f)
// c, p, l, i = the four reconstructions (B, 1, D, H, W), t* the ground truths, z* the latents.  Composed of torch operators and three
// BatchDiceLoss calls it is ~60 kernels between the forward and the backward of a step (0.3 ms of a 6.5 ms step).
// sums (replica rows of 16 doubles): 0 hinge(p, i), 1 hinge(p, c), 2-4 Dice(c), 5-7 Dice(p), 8-10 Dice(l), 11 sum |zi - zl|
__global__ __launch_bounds__(256) void cae_loss_sums_kernel(const float* __restrict__ c, int64_t cbs, const float* __restrict__ p, int64_t pbs,
                                                            const float* __restrict__ l, int64_t lbs, const float* __restrict__ ii, int64_t ibs,
                                                            const float* __restrict__ tc, int64_t tcbs, const float* __restrict__ tp, int64_t tpbs,
                                                            const float* __restrict__ tl, int64_t tlbs, int64_t DHW, double* __restrict__ sums,
                                                            int B, const float* __restrict__ zi, const float* __restrict__ zl, int64_t nlat) {
  __shared__ float red[4 * 11];      // [wave][sum], added up in wave order (sp_cols_sum)
  if ((int)blockIdx.y == B) {        // the latent term: sum |zi - zl| -> column 11
    float t = 0.f;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nlat; k += (int64_t)gridDim.x * 256) t += fabsf(zi[k] - zl[k]);
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[(size_t)(blockIdx.x % SP_REDUCE_ROWS) * 16 + 11], (double)((red[0] + red[1]) + (red[2] + red[3])));
    return;
  }
  const int b = blockIdx.y;
  c += b * cbs; p += b * pbs; l += b * lbs; ii += b * ibs; tc += b * tcbs; tp += b * tpbs; tl += b * tlbs;
  float s[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) s[k] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < DHW; v += (int64_t)gridDim.x * 256) {
    const float vc = c[v], vp = p[v], vl = l[v], vi = ii[v], a = tc[v], bq = tp[v], d = tl[v];
    const float d1 = vp - vi, d2 = vp - vc;
    s[0] += fabsf(d1) - d1; s[1] += fabsf(d2) - d2;
    s[2] += vc * a; s[3] += vc * vc; s[4] += a * a;
    s[5] += vp * bq; s[6] += vp * vp; s[7] += bq * bq;
    s[8] += vl * d; s[9] += vl * vl; s[10] += d * d;
  }
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 11 + k] = w;
  }
  __syncthreads();
  if (threadIdx.x < 11)
    atomicAdd(&sums[(size_t)((blockIdx.x + blockIdx.y) % SP_REDUCE_ROWS) * 16 + threadIdx.x], (double)sp_cols_sum(red, 11, 4, threadIdx.x));
}
// one thread: the loss and the backward's coefficients
//   coef: 0 hinge scale 1 / (N (5 + f)); (1, 2) (3, 4) (5, 6) Dice (ca, cb) / (5 + f) of c, p, l; 7 latent scale f / (nlat (5 + f))
__global__ void cae_loss_finalize_kernel(const double* __restrict__ sums, int64_t nlat, double N, float w, double eps, float factor,
                                         float* __restrict__ loss, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double lat = nlat > 0 ? sp_rows_sum(sums, 11, 16) / (double)nlat : 0.0;
  const double den0 = 5.0 + (double)factor;
  double acc = sp_rows_sum(sums, 0, 16) / N + sp_rows_sum(sums, 1, 16) / N;
  for (int k = 0; k < 3; ++k) {
    const double num = 2.0 * sp_rows_sum(sums, 2 + 3 * k, 16) + eps;
    const double den = sp_rows_sum(sums, 3 + 3 * k, 16) + sp_rows_sum(sums, 4 + 3 * k, 16) + eps;
    acc += 1.0 - (double)w * num / den;
    coef[1 + 2 * k] = (float)(-2.0 * w / den / den0);
    coef[2 + 2 * k] = (float)(2.0 * w * num / (den * den) / den0);
  }
  acc += (double)factor * lat;
  *loss = (float)(acc / den0);
  coef[0] = (float)(1.0 / (N * den0));
  coef[7] = nlat > 0 ? (float)((double)factor / ((double)nlat * den0)) : 0.f;
}
// gradients of the four reconstructions (dense (B, DHW) each, at dc / dp / dl / di) and of the two latents; up: dL/dloss on the device
__global__ __launch_bounds__(256) void cae_loss_bwd_kernel(const float* __restrict__ c, int64_t cbs, const float* __restrict__ p, int64_t pbs,
                                                           const float* __restrict__ l, int64_t lbs, const float* __restrict__ ii, int64_t ibs,
                                                           const float* __restrict__ tc, int64_t tcbs, const float* __restrict__ tp, int64_t tpbs,
                                                           const float* __restrict__ tl, int64_t tlbs, int64_t DHW, int B, const float* __restrict__ coef,
                                                           const float* __restrict__ up, float* __restrict__ dc, float* __restrict__ dp,
                                                           float* __restrict__ dl, float* __restrict__ di, const float* __restrict__ zi,
                                                           const float* __restrict__ zl, int64_t nlat, float* __restrict__ dzi, float* __restrict__ dzl) {
  const float go = up[0];
  if ((int)blockIdx.y == B) {      // the latents
    const float ls = coef[7] * go;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nlat; k += (int64_t)gridDim.x * 256) {
      const float d = zi[k] - zl[k];
      const float g = d > 0.f ? ls : (d < 0.f ? -ls : 0.f);
      dzi[k] = g; dzl[k] = -g;
    }
    return;
  }
  const int b = blockIdx.y;
  c += b * cbs; p += b * pbs; l += b * lbs; ii += b * ibs; tc += b * tcbs; tp += b * tpbs; tl += b * tlbs;
  dc += (int64_t)b * DHW; dp += (int64_t)b * DHW; dl += (int64_t)b * DHW; di += (int64_t)b * DHW;
  const float hs = coef[0] * go;
  const float cac = coef[1] * go, cbc = coef[2] * go, cap = coef[3] * go, cbp = coef[4] * go, cal = coef[5] * go, cbl = coef[6] * go;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < DHW; v += (int64_t)gridDim.x * 256) {
    const float vc = c[v], vp = p[v], vl = l[v], vi = ii[v];
    const float d1 = vp - vi, d2 = vp - vc;
    const float g1 = ((d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f)) - 1.f) * hs;      // d/dd (|d| - d), sign(0) = 0 as torch.abs
    const float g2 = ((d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f)) - 1.f) * hs;
    dp[v] = g1 + g2 + (cap * tp[v] + cbp * vp);
    di[v] = -g1;
    dc[v] = -g2 + (cac * tc[v] + cbc * vc);
    dl[v] = cal * tl[v] + cbl * vl;
  }
}
extern "C" int sp_cae_loss_fwd(const float* c, int64_t cbs, const float* p, int64_t pbs, const float* l, int64_t lbs, const float* i, int64_t ibs,
                               const float* tc, int64_t tcbs, const float* tp, int64_t tpbs, const float* tl, int64_t tlbs, int32_t B, int64_t DHW,
                               const float* zi, const float* zl, int64_t nlat, float dice_weight, double eps, float factor, double* sums,
                               float* loss, float* coef, sp_stream_t stream) {
  SP_CHECK_ARG(c && p && l && i && tc && tp && tl && sums && loss && coef && B >= 1 && B <= 65534 && DHW >= 1 && (nlat == 0 || (zi && zl)), "sp_cae_loss_fwd: bad arguments");
  int64_t gx = (DHW + 256 * 8 - 1) / (256 * 8);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(cae_loss_sums_kernel, dim3((unsigned)gx, B + (nlat > 0 ? 1 : 0)), dim3(256), 0, ST(stream), c, cbs, p, pbs, l, lbs, i, ibs, tc, tcbs, tp, tpbs, tl,
                     tlbs, DHW, sums, B, zi, zl, nlat);
  hipLaunchKernelGGL(cae_loss_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), sums, nlat, (double)B * (double)DHW, dice_weight, eps, factor, loss, coef);
  SP_CHECK_LAUNCH("sp_cae_loss_fwd");
  return SP_OK;
}
extern "C" int sp_cae_loss_bwd(const float* c, int64_t cbs, const float* p, int64_t pbs, const float* l, int64_t lbs, const float* i, int64_t ibs,
                               const float* tc, int64_t tcbs, const float* tp, int64_t tpbs, const float* tl, int64_t tlbs, int32_t B, int64_t DHW,
                               const float* coef, const float* up, float* dc, float* dp, float* dl, float* di, const float* zi, const float* zl,
                               int64_t nlat, float* dzi, float* dzl, sp_stream_t stream) {
  SP_CHECK_ARG(c && p && l && i && tc && tp && tl && coef && up && dc && dp && dl && di && B >= 1 && B <= 65534 && DHW >= 1 && (nlat == 0 || (zi && zl && dzi && dzl)),
               "sp_cae_loss_bwd: bad arguments");
  int64_t gx = (DHW + 256 * 8 - 1) / (256 * 8);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(cae_loss_bwd_kernel, dim3((unsigned)gx, B + (nlat > 0 ? 1 : 0)), dim3(256), 0, ST(stream), c, cbs, p, pbs, l, lbs, i, ibs, tc, tcbs, tp, tpbs,
                     tl, tlbs, DHW, B, coef, up, dc, dp, dl, di, zi, zl, nlat, dzi, dzl);
  SP_CHECK_LAUNCH("sp_cae_loss_bwd");
  return SP_OK;
}
// do[b,c,v] = up * (ca[c]*t + cb[c]*o), up = *upstream (the scalar gradient of the loss, read on the device)
__global__ void dice_bwd_kernel(const float* __restrict__ o, int64_t obs, const float* __restrict__ t, int64_t tbs,
                                const float* __restrict__ coef, const float* __restrict__ upstream, int C, int64_t DHW,
                                int64_t total, float* __restrict__ d) {
  const float up = upstream ? *upstream : 1.f;
  const int64_t per_b = (int64_t)C * DHW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per_b, r = i - b * per_b;
    const int c = (int)(r / DHW);
    d[i] = up * (coef[2 * c] * t[b * tbs + r] + coef[2 * c + 1] * o[b * obs + r]);
  }
}
extern "C" int sp_dice_bwd(const float* o, int64_t o_bstride, const float* t, int64_t t_bstride, const float* coef,
                           const float* upstream, int32_t B, int32_t C, int64_t DHW, float* dout, sp_stream_t stream) {
  SP_CHECK_ARG(o && t && coef && dout, "sp_dice_bwd: null pointer");
  const int64_t total = (int64_t)B * C * DHW;
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(dice_bwd_kernel, dim3(grid), dim3(256), 0, ST(stream), o, o_bstride, t, t_bstride, coef, upstream, C, DHW,
                     total, dout);
  SP_CHECK_LAUNCH("sp_dice_bwd");
  return SP_OK;
}

// ------------------------------------------------------------------------------------------------ utilities
// tp / fp / fn / tn of (result > thr) against (target > thr) -- the counts behind Dice / precision / sensitivity /
// specificity of the reference's batch metrics (metrics.py:31-62), without copying the volumes to the host
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ r, const float* __restrict__ t, float thr,
                                                         int64_t n, unsigned long long* __restrict__ counts) {
  unsigned int c[4] = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bool a = r[i] > thr, b = t[i] > thr;
    c[0] += a && b; c[1] += a && !b; c[2] += !a && b; c[3] += !a && !b;
  }
  __shared__ unsigned int red[4];
  if (threadIdx.x < 4) red[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned int v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(&counts[threadIdx.x], (unsigned long long)red[threadIdx.x]);
}
extern "C" int sp_confusion_counts(const float* result, const float* target, float threshold, int64_t n,
                                   unsigned long long* counts, sp_stream_t stream) {
  SP_CHECK_ARG(result && target && counts && n >= 0, "sp_confusion_counts: bad arguments");
  if (n == 0) return SP_OK;
  const unsigned grid = (unsigned)((n + 256 * 16 - 1) / (256 * 16) > 1024 ? 1024 : (n + 256 * 16 - 1) / (256 * 16));
  hipLaunchKernelGGL(confusion_kernel, dim3(grid), dim3(256), 0, ST(stream), result, target, threshold, n, counts);
  SP_CHECK_LAUNCH("sp_confusion_counts");
  return SP_OK;
}

__global__ void add_f64_to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] += scale * (float)src[i];
}
extern "C" int sp_add_f64_to_f32(const double* src, float* dst, int64_t n, float scale, sp_stream_t stream) {
  SP_CHECK_ARG(src && dst && n > 0, "sp_add_f64_to_f32: bad arguments");
  hipLaunchKernelGGL(add_f64_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), src, dst, n, scale);
  SP_CHECK_LAUNCH("sp_add_f64_to_f32");
  return SP_OK;
}

// out[b,i] = c[b,i] + step[b]*(p[b,i]-c[b,i])   (Enc3D._interpolate, Cae3D.py:78-89)
template <typename T>
__global__ void lerp_batch_kernel(const T* __restrict__ c, const T* __restrict__ p, const float* __restrict__ step,
                                  T* __restrict__ out, int64_t per_b, int64_t total8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 8;
    const float s = step[e / per_b];
    float a[8], b[8];
    Store<T>::ld8(c + e, a);
    Store<T>::ld8(p + e, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = a[j] + s * (b[j] - a[j]);
    Store<T>::st8(out + e, a);
  }
}
extern "C" int sp_lerp_batch(const void* c, const void* p, const float* step, void* out, int32_t dtype, int32_t B,
                             int64_t per_b, sp_stream_t stream) {
  SP_CHECK_ARG(c && p && step && out && per_b % 8 == 0, "sp_lerp_batch: bad arguments");
  const int64_t total8 = (int64_t)B * per_b / 8;
  const unsigned grid = (unsigned)((total8 + 255) / 256 > 2048 ? 2048 : (total8 + 255) / 256);
  if (dtype == SP_BF16) hipLaunchKernelGGL(lerp_batch_kernel<bf16_t>, dim3(grid), dim3(256), 0, ST(stream), (const bf16_t*)c, (const bf16_t*)p, step, (bf16_t*)out, per_b, total8);
  else hipLaunchKernelGGL(lerp_batch_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)c, (const float*)p, step, (float*)out, per_b, total8);
  SP_CHECK_LAUNCH("sp_lerp_batch");
  return SP_OK;
}

// out = a*x + b*y elementwise on channels-last buffers (n multiple of 8)
template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, int64_t n8, float a, float b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float u[8], v[8];
    Store<T>::ld8(x + i * 8, u);
    Store<T>::ld8(y + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = a * u[j] + b * v[j];
    Store<T>::st8(out + i * 8, u);
  }
}
extern "C" int sp_axpby(const void* x, const void* y, void* out, int32_t dtype, int64_t n, float a, float b, sp_stream_t stream) {
  SP_CHECK_ARG(x && y && out && n % 8 == 0, "sp_axpby: bad arguments");
  const int64_t n8 = n / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
  if (dtype == SP_BF16) hipLaunchKernelGGL(axpby_kernel<bf16_t>, dim3(grid), dim3(256), 0, ST(stream), (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)out, n8, a, b);
  else hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)x, (const float*)y, (float*)out, n8, a, b);
  SP_CHECK_LAUNCH("sp_axpby");
  return SP_OK;
}

// ------------------------------------------------------------------------------------------------ Adam
// torch.optim.Adam (no amsgrad): g += wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr_over_bc1, float beta1, float beta2, float eps, float wd, float inv_sqrt_bc2,
                            float grad_scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pi = p[i];
    const float gi = g[i] * grad_scale + wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}
// graph-capturable form: the step count lives in device memory (incremented by the caller, on the stream,
// before this launch); bias corrections are formed on the device.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                                const int32_t* __restrict__ step_ptr, float grad_scale) {
  const float step = (float)(*step_ptr);
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float lr_over_bc1 = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pi = p[i];
    const float gi = g[i] * grad_scale + wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}
extern "C" int sp_adam_step_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, const int32_t* step_dev,
                                     float grad_scale, sp_stream_t stream) {
  SP_CHECK_ARG(p && g && m && v && n > 0 && step_dev, "sp_adam_step_flat_dev: bad arguments");
  const unsigned grid = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, step_dev, grad_scale);
  SP_CHECK_LAUNCH("sp_adam_step_flat_dev");
  return SP_OK;
}

// hyper-parameters in device memory as well ({lr, beta1, beta2, eps, weight_decay}): a captured step keeps following
// Learner.adapt_lr / adapt_betas (Learner.py:156-161, CaeReconstructionLearner.py:28-40) -- the host rewrites the five
// floats between replays instead of re-capturing.
__global__ void adam_hyp_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                int64_t n, const float* __restrict__ hyper, const int32_t* __restrict__ step_ptr, float grad_scale) {
  const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const float step = (float)(*step_ptr);
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float lr_over_bc1 = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pi = p[i];
    const float gi = g[i] * grad_scale + wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}
extern "C" int sp_adam_step_flat_hyp(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                                     const int32_t* step_dev, float grad_scale, sp_stream_t stream) {
  SP_CHECK_ARG(p && g && m && v && n > 0 && step_dev && hyper_dev, "sp_adam_step_flat_hyp: bad arguments");
  const unsigned grid = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_hyp_kernel, dim3(grid), dim3(256), 0, ST(stream), p, g, m, v, n, hyper_dev, step_dev, grad_scale);
  SP_CHECK_LAUNCH("sp_adam_step_flat_hyp");
  return SP_OK;
}

extern "C" int sp_adam_step_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                                 sp_stream_t stream) {
  SP_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "sp_adam_step_flat: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const unsigned grid = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, ST(stream), p, g, m, v, n, (float)(lr / bc1), beta1, beta2, eps,
                     weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale);
  SP_CHECK_LAUNCH("sp_adam_step_flat");
  return SP_OK;
}
