// Table-driven implicit-GEMM 3-D convolution for gfx950 (MFMA 16x16x32 bf16).
//
// Replaces the ATen/cuDNN kernels behind nn.Conv3d / nn.ConvTranspose3d at
// Unet3D.py:19,22 and Cae3D.py:41-74,178-218 (forward) and their data gradients.
//
// GEMM view per workgroup:  D[cout][voxel] += W[cout][k] * X[k][voxel],  k = (tap, cin).
//   * A operand = weights, pre-packed per K step as MFMA fragments (sp_conv_prep_weights), read
//     straight from L2 (every workgroup reads the same few KB);
//   * B operand = activations: the input halo tile is staged ONCE into LDS (BatchNorm applied on
//     load, zero padding after the norm), in 16-channel planes [plane][voxel][16] so that the
//     per-lane ds_read_b128 of one (tap, 8-channel octet) is bank-conflict free without padding;
//   * a K step (32) = 4 octets, one per 16-lane group; which (tap, octet) each group reads is a
//     host-built table of LDS byte offsets (ktab) -> stride, padding, tap subsets (transposed
//     convolution parity classes) and channel chunking need no kernel variants;
//   * D lands as 4 consecutive output channels of one voxel per lane -> 8/16-byte stores into
//     the channels-last output, plus fused bias, activation and per-channel sum / sum-of-squares
//     (the next BatchNorm's batch statistics).
//   * SP_F32 mode: x = hi + lo in bf16, three MFMAs per product (hi*hi, hi*lo, lo*hi), fp32
//     accumulate: ~2^-17 relative, used for parity against the fp32 CPU reference.
#include "sp_common.h"

struct ConvDev {
  sp_conv_args a;
  FastDiv d_octs, d_itw, d_ith;   // staging index math
  FastDiv d_tx, d_ty, d_tz;       // block id -> tile
  uint32_t ntx, nty, ntz, nblk;
};

template <int NT, int MT, int NP, typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvDev P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_conv_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lv = lane & 15, lg = lane >> 4;

  // ---- which tile --------------------------------------------------------------------------
  uint32_t bid = xcd_remap(blockIdx.x, P.nblk);
  uint32_t t = bid;
  uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
  q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; t = q;
  q = fdiv(t, P.d_tz); const int tz = t - q * P.ntz; const int b = q;
  const int oz0 = tz * a.TD, oy0 = ty * a.TH, ox0 = tx * 16;
  const int nt0 = blockIdx.y * NT;
  // input coordinate of LDS tile voxel (0,0,0)
  const int iz0 = oz0 * a.sD + a.o0D, iy0 = oy0 * a.sH + a.o0H, ix0 = ox0 * a.sW + a.o0W;

  int* ktab_l = reinterpret_cast<int*>(lds);
  const int ktab_bytes = (a.steps_per_group * 16 + 15) & ~15;
  unsigned char* tile = lds + ktab_bytes;
  for (int i = tid; i < a.steps_per_group * 4; i += 256) ktab_l[i] = a.ktab[i];

  // per-lane LDS base of each M tile (row of 16 output voxels)
  int vbase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = wave * MT + m;
    const int rz = r / a.TH, ry = r - rz * a.TH;
    vbase[m] = ((rz * a.sD * a.ITH + ry * a.sH) * a.ITW + lv * a.sW) * a.vsb;
  }

  f32x4 acc[NT][MT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const TIN* __restrict__ xin = reinterpret_cast<const TIN*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * a.CPi;
  const bf16x8* __restrict__ wf_hi = reinterpret_cast<const bf16x8*>(a.wfrag_hi);
  const bf16x8* __restrict__ wf_lo = reinterpret_cast<const bf16x8*>(a.wfrag_lo);
  const int nvox_tile = a.ITD * a.ITH * a.ITW;
  const int nchunks = nvox_tile * a.octs_per_group;

  for (int grp = 0; grp < a.ngroups; ++grp) {
    if (grp > 0) __syncthreads();   // previous group's reads are done
    // ---- stage the halo tile: global (channels-last) -> norm -> bf16 (hi/lo) -> LDS planes ----
    const int oct0 = grp * a.octs_per_group;
    for (int i = tid; i < nchunks; i += 256) {
      const uint32_t vox = fdiv(i, P.d_octs);
      const int oc = i - vox * a.octs_per_group;
      const uint32_t row = fdiv(vox, P.d_itw);
      const int vx = vox - row * a.ITW;
      const uint32_t vz = fdiv(row, P.d_ith);
      const int vy = row - vz * a.ITH;
      const int gz = iz0 + (int)vz, gy = iy0 + vy, gx = ix0 + vx;
      float v[8];
      const bool inb = (unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi;
      if (inb) {
        const int c = (oct0 + oc) * 8;
        Store<TIN>::ld8(xin + (((size_t)gz * a.Hi + gy) * a.Wi + gx) * a.CPi + c, v);
        if (a.in_scale) {
          const float4 s0 = *reinterpret_cast<const float4*>(a.in_scale + c), s1 = *reinterpret_cast<const float4*>(a.in_scale + c + 4);
          const float4 h0 = *reinterpret_cast<const float4*>(a.in_shift + c), h1 = *reinterpret_cast<const float4*>(a.in_shift + c + 4);
          v[0] = fmaf(v[0], s0.x, h0.x); v[1] = fmaf(v[1], s0.y, h0.y); v[2] = fmaf(v[2], s0.z, h0.z); v[3] = fmaf(v[3], s0.w, h0.w);
          v[4] = fmaf(v[4], s1.x, h1.x); v[5] = fmaf(v[5], s1.y, h1.y); v[6] = fmaf(v[6], s1.z, h1.z); v[7] = fmaf(v[7], s1.w, h1.w);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      const int pl = oc / a.opp, po = oc - pl * a.opp;
      unsigned char* dst = tile + pl * a.plane_bytes + vox * a.vsb + po * 16;
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (uint32_t)f2bf(v[2 * j]) | ((uint32_t)f2bf(v[2 * j + 1]) << 16);
      *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      if (NP == 2) {
        uint32_t wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float r0 = v[2 * j] - __uint_as_float(w[j] << 16);
          const float r1 = v[2 * j + 1] - __uint_as_float(w[j] & 0xffff0000u);
          wl[j] = (uint32_t)f2bf(r0) | ((uint32_t)f2bf(r1) << 16);
        }
        *reinterpret_cast<uint4*>(dst + a.lo_offset) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
      }
    }
    __syncthreads();

    // ---- K loop: taps x channel octets of this group ------------------------------------------
    const size_t gstep0 = (size_t)grp * a.steps_per_group;
    for (int s = 0; s < a.steps_per_group; ++s) {
      const int koff = ktab_l[s * 4 + lg];
      const size_t fbase = ((gstep0 + s) * a.NTtot + nt0) * 64 + lane;
      bf16x8 wa[NT], wl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        wa[n] = wf_hi[fbase + (size_t)n * 64];
        if (NP == 2) wl[n] = wf_lo[fbase + (size_t)n * 64];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(tile + vbase[m] + koff);
        bf16x8 xl;
        if (NP == 2) xl = *reinterpret_cast<const bf16x8*>(tile + a.lo_offset + vbase[m] + koff);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[n], xb, acc[n][m], 0, 0, 0);
          if (NP == 2) {
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[n], xl, acc[n][m], 0, 0, 0);
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], xb, acc[n][m], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: bias, activation, statistics, channels-last store ----------------------------
  TOUT* __restrict__ yout = reinterpret_cast<TOUT*>(a.y) + (size_t)b * a.YD * a.YH * a.YW * a.CPo;
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[n][j] = s2[n][j] = 0.f;

  const int ox = ox0 + lv;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int c0 = (nt0 + n) * 16 + lg * 4;
    float bj[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) { const float4 bb = *reinterpret_cast<const float4*>(a.bias + c0); bj[0] = bb.x; bj[1] = bb.y; bj[2] = bb.z; bj[3] = bb.w; }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int r = wave * MT + m;
      const int rz = r / a.TH, ry = r - rz * a.TH;
      const int oz = oz0 + rz, oy = oy0 + ry;
      const bool valid = oz < a.Do && oy < a.Ho && ox < a.Wo;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = acc[n][m][j] + bj[j];
        z = act_fwd(a.act, a.act_param, z);
        v[j] = (c0 + j < a.Cout) ? z : 0.f;
      }
      if (valid && c0 < a.CPo) {
        if (sizeof(TOUT) == 2) {   // statistics of what is actually stored (bf16-rounded values)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = bf2f(f2bf(v[j]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[n][j] += v[j]; s2[n][j] += v[j] * v[j]; }
        const size_t off = (((size_t)(oz * a.osD + a.ooD) * a.YH + (oy * a.osH + a.ooH)) * a.YW + (ox * a.osW + a.ooW)) * a.CPo + c0;
        Store<TOUT>::st4(yout + off, v);
      }
    }
  }

  if (a.stats) {
    __syncthreads();                       // tile no longer needed: reuse LDS for the block reduction
    float* red = reinterpret_cast<float*>(lds);
    for (int i = tid; i < NT * 16 * 2; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x1 = s1[n][j], x2 = s2[n][j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
        if (lv == 0) {
          atomicAdd(&red[(n * 16 + lg * 4 + j) * 2], x1);
          atomicAdd(&red[(n * 16 + lg * 4 + j) * 2 + 1], x2);
        }
      }
    __syncthreads();
    for (int i = tid; i < NT * 16 * 2; i += 256) {
      const int c = nt0 * 16 + (i >> 1);
      if (c < a.CPo) atomicAdd(&a.stats[(size_t)c * 2 + (i & 1)], (double)red[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int NT, int MT, int NP, typename TIN, typename TOUT>
static int launch_conv(const ConvDev& P, dim3 grid, hipStream_t st) {
  auto kern = conv_igemm_kernel<NT, MT, NP, TIN, TOUT>;
  if (P.a.lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P.a.lds_bytes);
    if (e != hipSuccess) { sp_set_error("sp_conv3d_igemm: cannot raise LDS limit to %d: %s", P.a.lds_bytes, hipGetErrorString(e)); return SP_EHIP; }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), P.a.lds_bytes, st, P);
  SP_CHECK_LAUNCH("sp_conv3d_igemm");
  return SP_OK;
}

template <int NT, int MT>
static int dispatch_dtype(const ConvDev& P, dim3 grid, hipStream_t st) {
  const int di = P.a.dtype_in, dout = P.a.dtype_out;
  if (di == SP_BF16 && dout == SP_BF16) return launch_conv<NT, MT, 1, bf16_t, bf16_t>(P, grid, st);
  if (di == SP_F32 && dout == SP_F32) return launch_conv<NT, MT, 2, float, float>(P, grid, st);
  if (di == SP_BF16 && dout == SP_F32) return launch_conv<NT, MT, 1, bf16_t, float>(P, grid, st);
  sp_set_error("sp_conv3d_igemm: unsupported dtype pair in=%d out=%d", di, dout);
  return SP_EINVAL;
}

extern "C" int sp_conv3d_igemm(const sp_conv_args* a, sp_stream_t stream) {
  SP_CHECK_ARG(a && a->x && a->y && a->wfrag_hi && a->ktab, "sp_conv3d_igemm: null pointer");
  SP_CHECK_ARG(a->CPi % 8 == 0 && a->CPo % 8 == 0, "sp_conv3d_igemm: channel pitch must be a multiple of 8 (CPi=%d CPo=%d)", a->CPi, a->CPo);
  SP_CHECK_ARG(a->TD * a->TH == 4 * a->MT, "sp_conv3d_igemm: TD*TH (%d*%d) must equal 4*MT (%d)", a->TD, a->TH, 4 * a->MT);
  SP_CHECK_ARG(a->NTtot % a->NT == 0, "sp_conv3d_igemm: NTtot %d not a multiple of NT %d", a->NTtot, a->NT);
  SP_CHECK_ARG(a->ngroups * a->octs_per_group * 8 == a->CPi, "sp_conv3d_igemm: groups (%d x %d octets) do not cover CPi=%d", a->ngroups, a->octs_per_group, a->CPi);
  SP_CHECK_ARG(a->dtype_in != SP_F32 || (a->wfrag_lo && a->lo_offset > 0), "sp_conv3d_igemm: f32 mode needs wfrag_lo and lo_offset");
  SP_CHECK_ARG(a->Do > 0 && a->Ho > 0 && a->Wo > 0 && a->B > 0, "sp_conv3d_igemm: empty output");
  // the staged tile must cover every tap of every output row of the tile
  SP_CHECK_ARG(a->ITW >= 15 * a->sW + 1 && a->ITH >= (a->TH - 1) * a->sH + 1 && a->ITD >= (a->TD - 1) * a->sD + 1, "sp_conv3d_igemm: input tile smaller than output tile");
  {
    const int planes = (a->octs_per_group + a->opp - 1) / a->opp;
    const long tile_bytes = (long)planes * a->plane_bytes * (a->dtype_in == SP_F32 ? 2 : 1);
    const long need = ((a->steps_per_group * 16 + 15) & ~15) + tile_bytes;
    SP_CHECK_ARG(a->plane_bytes >= a->ITD * a->ITH * a->ITW * a->vsb && need <= a->lds_bytes && a->lds_bytes <= 160 * 1024,
                 "sp_conv3d_igemm: LDS plan inconsistent (need %ld, lds_bytes %d)", need, a->lds_bytes);
    SP_CHECK_ARG(a->lds_bytes >= a->NT * 16 * 2 * 4, "sp_conv3d_igemm: LDS too small for the reduction");
  }
  ConvDev P;
  P.a = *a;
  P.d_octs = make_fastdiv(a->octs_per_group);
  P.d_itw = make_fastdiv(a->ITW);
  P.d_ith = make_fastdiv(a->ITH);
  P.ntx = (a->Wo + 15) / 16;
  P.nty = (a->Ho + a->TH - 1) / a->TH;
  P.ntz = (a->Do + a->TD - 1) / a->TD;
  P.d_tx = make_fastdiv(P.ntx);
  P.d_ty = make_fastdiv(P.nty);
  P.d_tz = make_fastdiv(P.ntz);
  const uint64_t nblk = (uint64_t)P.ntx * P.nty * P.ntz * a->B;
  SP_CHECK_ARG(nblk < (1ull << 31), "sp_conv3d_igemm: grid too large");
  P.nblk = (uint32_t)nblk;
  dim3 grid(P.nblk, a->NTtot / a->NT);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define SP_CASE(NT_, MT_) if (a->NT == NT_ && a->MT == MT_) return dispatch_dtype<NT_, MT_>(P, grid, st)
  SP_CASE(1, 8); SP_CASE(2, 8); SP_CASE(4, 8);
  SP_CASE(1, 4); SP_CASE(2, 4); SP_CASE(4, 4);
  SP_CASE(1, 2); SP_CASE(2, 2); SP_CASE(4, 2);
#undef SP_CASE
  sp_set_error("sp_conv3d_igemm: no kernel for NT=%d MT=%d", a->NT, a->MT);
  return SP_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// weight re-packing: fp32 (Cout,Cin,taps) -> MFMA A fragments [step][ntile][lane][8] (hi / lo bf16)
__global__ void prep_wfrag_kernel(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin,
                                  const int32_t* __restrict__ kmap, int nsteps, int NTtot,
                                  bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (step, ntile, lane)
  const int64_t total = (int64_t)nsteps * NTtot * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int nt = (idx >> 6) % NTtot;
  const int s = (idx >> 6) / NTtot;
  const int co = nt * 16 + (lane & 15), g = lane >> 4;
  const int km = kmap[s * 4 + g];
  uint32_t wh[4], wl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float f[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v = 0.f;
      if (km >= 0) {
        const int tap = km >> 16, ci = (km & 0xffff) * 8 + 2 * j + h;
        if (co < Cout && ci < Cin) v = w[co * sCo + ci * sCi + tap];
      }
      f[h] = v;
    }
    const bf16_t h0 = f2bf(f[0]), h1 = f2bf(f[1]);
    wh[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    wl[j] = (uint32_t)f2bf(f[0] - bf2f(h0)) | ((uint32_t)f2bf(f[1] - bf2f(h1)) << 16);
  }
  reinterpret_cast<uint4*>(hi)[idx] = make_uint4(wh[0], wh[1], wh[2], wh[3]);
  if (lo) reinterpret_cast<uint4*>(lo)[idx] = make_uint4(wl[0], wl[1], wl[2], wl[3]);
}

extern "C" int sp_conv_prep_weights(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin,
                                    const int32_t* kmap, int32_t nsteps, int32_t NTtot, void* wfrag_hi,
                                    void* wfrag_lo, sp_stream_t stream) {
  SP_CHECK_ARG(w && kmap && wfrag_hi && nsteps > 0 && NTtot > 0, "sp_conv_prep_weights: bad arguments");
  const int64_t total = (int64_t)nsteps * NTtot * 64;
  hipLaunchKernelGGL(prep_wfrag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, sCo, sCi, Cout, Cin, kmap, nsteps, NTtot,
                     reinterpret_cast<bf16_t*>(wfrag_hi), reinterpret_cast<bf16_t*>(wfrag_lo));
  SP_CHECK_LAUNCH("sp_conv_prep_weights");
  return SP_OK;
}
