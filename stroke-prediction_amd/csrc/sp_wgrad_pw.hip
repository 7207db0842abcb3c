// Weight gradient of the pointwise (1x1x1, stride 1) convolutions -- the CAE's tail (Cae3D.py:214-218: 16 -> 16 and 16 -> 1 at
// 28 x 128 x 128) and generic classify heads (32 -> 32, 32 -> 2 of the 4-scale net): dw[co][ci] = sum_v dz[v][co] * x[v][ci], a
// reduction over millions of voxels with a 16..32-wide result, HBM-bound by construction (read both tensors once).  The
// register-staged generic kernel needed 85 us for 117 MB (16 -> 16) and 580 us for 1.1 GB (32 -> 32 @164^3).
//   * a WAVE streams 32-voxel chunks of both operands into its own two LDS buffers with LDS-DMA (one instruction per
//     16-channel plane and chunk, the next chunk in flight while this one is used; no workgroup barrier in the loop);
//   * both MFMA operands are read back transposed (ds_read_b64_tr_b16: K = 32 voxels), COT x CIT MFMAs per chunk;
//   * the four waves' sums are added through LDS and written as the workgroup's partial block [CoP][CiP] (parts mode), which
//     sp_wgrad_finish_folded sums -- that kernel also applies the BatchNorm in front of the layer (dw = s*acc + t*sum dz).
#include "sp_common.h"

__device__ uint4 sp_pw_zero_page[64];

struct WgradPwDev {
  sp_wgrad_args a;
  int64_t M;          // voxels
};

__device__ __forceinline__ bf16x8 pw_tr_read2(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p0));
  bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p1));
  return __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int COT, int CIT>
__global__ __launch_bounds__(256) void wgrad_pw_kernel(const WgradPwDev P) {
  constexpr int NPL = COT + CIT;                    // 16-channel planes staged per chunk (1 KiB each)
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 2 * NPL * 1024];
  const sp_wgrad_args& a = P.a;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int co_t0 = blockIdx.y * COT, ci_t0 = blockIdx.z * CIT;
  unsigned char* mybuf = lds + wave * (2 * NPL * 1024);
  // transposed-read lane offsets inside a [32 voxels][32 bytes] plane (same voxel permutation for both operands)
  const int vq0 = ((lg & 1) ? 2 * lg + 1 : 2 * lg) * 4 + lq;
  const int vq1 = ((lg & 1) ? 2 * lg : 2 * lg + 1) * 4 + lq;
  const int off0 = vq0 * 32 + lp * 8, off1 = vq1 * 32 + lp * 8;
  const int dv = lane >> 1, dh = lane & 1;          // DMA role: voxel of the chunk, channel half of the plane
  const unsigned char* zeros = reinterpret_cast<const unsigned char*>(sp_pw_zero_page) + lane * 16;
  const unsigned char* xg = reinterpret_cast<const unsigned char*>(a.x);
  const unsigned char* dg = reinterpret_cast<const unsigned char*>(a.dz);
  f32x4 acc[COT][CIT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int i = 0; i < CIT; ++i) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // BatchNorm groups (a.groups > 1; host: nblocks % groups == 0, voxels per group a multiple of 32): a workgroup's chunks stay inside
  // its group, so the partial blocks of a group are consecutive (sp_wgrad_finish_folded_groups)
  const int ngr = a.groups > 1 ? a.groups : 1;
  const int64_t ncg = (P.M + 31) / 32 / ngr;                 // chunks per group (one group: all)
  const int bpg = gridDim.x / ngr, grp = blockIdx.x / bpg, lb = blockIdx.x - grp * bpg;
  const int64_t nchunk = ngr > 1 ? (grp + 1) * ncg : (P.M + 31) / 32;
  const int64_t stride = (int64_t)bpg * 4;
  auto issue = [&](int64_t chunk, int buf) {
    const int64_t v = chunk * 32 + dv;
    const bool ok = chunk < nchunk && v < P.M;
    unsigned char* dst = mybuf + buf * (NPL * 1024);
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const bool okc = ok && (co_t0 + c) * 16 + dh * 8 < a.CPo;       // (a pitch of 8: the upper half of the tile reads zeros)
      sp_dma16_nc(okc ? dg + (v * a.CPo + (co_t0 + c) * 16 + dh * 8) * 2 : zeros, dst + c * 1024);
    }
#pragma unroll
    for (int i = 0; i < CIT; ++i) {
      const bool oki = ok && (ci_t0 + i) * 16 + dh * 8 < a.CPi;
      sp_dma16_nc(oki ? xg + (v * a.CPi + (ci_t0 + i) * 16 + dh * 8) * 2 : zeros, dst + (COT + i) * 1024);
    }
  };
  int64_t chunk = (ngr > 1 ? grp * ncg : (int64_t)0) + (int64_t)lb * 4 + wave;
  issue(chunk, 0);
  int buf = 0;
  for (; chunk < nchunk; chunk += stride) {
    issue(chunk + stride, buf ^ 1);                            // (past the end: zero page, never used)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL) : "memory");  // this chunk landed; the next one may be in flight
    const unsigned char* b = mybuf + buf * (NPL * 1024);
    bf16x8 af[COT], bf[CIT];
#pragma unroll
    for (int c = 0; c < COT; ++c) af[c] = pw_tr_read2(b + c * 1024 + off0, b + c * 1024 + off1);
#pragma unroll
    for (int i = 0; i < CIT; ++i) bf[i] = pw_tr_read2(b + (COT + i) * 1024 + off0, b + (COT + i) * 1024 + off1);
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int i = 0; i < CIT; ++i) acc[c][i] = SP_MFMA16(af[c], bf[i], acc[c][i], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads of this buffer are done before it is refilled
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- add the four waves through LDS, write the workgroup's partial block
  float* red = reinterpret_cast<float*>(lds);                   // [COT][CIT][256]
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float* d = red + (c * CIT + i) * 256 + (lg * 4 + j) * 16 + li;
            *d = w == 0 ? acc[c][i][j] : *d + acc[c][i][j];
          }
    }
    __syncthreads();
  }
  const int CoP = a.CoT * 16, CiP = a.CiT * 16;
  float* prow = a.dw_acc + (size_t)blockIdx.x * CoP * CiP;
  for (int e = threadIdx.x; e < COT * CIT * 256; e += 256) {
    const int t = e >> 8, r = e & 255, c = t / CIT, i = t - c * CIT;
    const int cot = co_t0 + c, cit = ci_t0 + i;
    if (cot < a.CoT && cit < a.CiT) prow[(size_t)(cot * 16 + (r >> 4)) * CiP + cit * 16 + (r & 15)] = red[e];
  }
}

// returns 1 when the layer is not a pointwise one this kernel handles (the caller falls back to the generic kernel)
int sp_wgrad_pw_try(const sp_wgrad_args* a, hipStream_t st) {
  const char* knob = getenv("SP_WGRAD_PW");
  if ((knob && atoi(knob) == 0) || !a->parts || a->dtype != SP_BF16 || a->in_scale || a->dz_scale) return 1;
  if (a->ntap != 1 || a->kD != 1 || a->kH != 1 || a->kW != 1 || a->sD != 1 || a->sH != 1 || a->sW != 1) return 1;
  if (a->o0D || a->o0H || a->o0W || a->Di != a->Do || a->Hi != a->Ho || a->Wi != a->Wo || a->x_plane) return 1;
  if (a->CPi % 8 || a->CPo % 8 || a->CiT != (a->CPi + 15) / 16 || a->CoT != (a->CPo + 15) / 16) return 1;
  WgradPwDev P;
  P.a = *a;
  P.M = (int64_t)a->B * a->Do * a->Ho * a->Wo;
  if (a->groups > 1) {
    SP_CHECK_ARG(a->nblocks % a->groups == 0 && a->B % a->groups == 0 && (P.M / a->groups) % 32 == 0,
                 "sp_conv3d_wgrad(pointwise): %d groups need nblocks %d %% groups == 0 and a multiple of 32 voxels per group", a->groups, a->nblocks);
  }
  const int COT = (a->CoT % 2 == 0) ? 2 : 1, CIT = (a->CiT % 2 == 0) ? 2 : 1;
  dim3 grid(a->nblocks, a->CoT / COT, a->CiT / CIT);
#define PW_CASE(C_, I_)                                                                       \
  if (COT == C_ && CIT == I_) {                                                               \
    hipLaunchKernelGGL((wgrad_pw_kernel<C_, I_>), grid, dim3(256), 0, st, P);                 \
    SP_CHECK_LAUNCH("sp_conv3d_wgrad(pointwise)");                                            \
    return SP_OK;                                                                             \
  }
  PW_CASE(1, 1) PW_CASE(1, 2) PW_CASE(2, 1) PW_CASE(2, 2)
#undef PW_CASE
  return 1;
}
