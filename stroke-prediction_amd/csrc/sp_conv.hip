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
#include <type_traits>

// one 8-channel chunk as loaded from global memory, kept raw until every load of a batch is in flight
template <typename T> struct RawChunk;
template <> struct RawChunk<bf16_t> {
  uint4 r;
  __device__ __forceinline__ void ld(const bf16_t* p, int64_t = 0) { r = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void unpack(float* v) const {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = sp_h2f_lo(w[i]); v[2 * i + 1] = sp_h2f_hi(w[i]); }
  }
};
template <> struct RawChunk<float> {
  float4 a, b;
  __device__ __forceinline__ void ld(const float* p, int64_t = 0) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void unpack(float* v) const {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
};
// bf16 pair (SP_HL): value = hi + lo, exact in fp32; the staging below splits it again (the same two words come back)
template <> struct RawChunk<sp_hl_t> {
  uint4 h, l;
  __device__ __forceinline__ void ld(const sp_hl_t* p, int64_t lo_delta) {
    h = *reinterpret_cast<const uint4*>(p);
    l = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p) + lo_delta);
  }
  __device__ __forceinline__ void unpack(float* v) const {
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = sp_h2f_lo(hw[i]) + sp_h2f_lo(lw[i]); v[2 * i + 1] = sp_h2f_hi(hw[i]) + sp_h2f_hi(lw[i]); }
  }
};
#define SB 8   // staging loads in flight per thread

// Diagnostic build only (-DSP_CONV_STAMPS): per-workgroup phase time stamps (shader clock) go to a buffer
// that nothing else reads; the shipped library compiles this away.
#ifdef SP_CONV_STAMPS
#define SP_NSTAMP 6
__device__ unsigned long long sp_stamp_buf[32768][SP_NSTAMP];   // shared with sp_conv_dma.hip (-fgpu-rdc in the stamp build)
#define STAMP(k)                                                                              \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    unsigned long long t_;                                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 32768) sp_stamp_buf[blockIdx.x][k] = t_; \
  } while (0)
extern "C" int sp_debug_read_stamps(unsigned long long* host, int nblocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sp_stamp_buf), sizeof(unsigned long long) * SP_NSTAMP * nblocks);
}
#else
#define STAMP(k)
#endif

struct ConvDev {
  sp_conv_args a;
  FastDiv d_octs, d_itw, d_ith;   // staging index math
  FastDiv d_tx, d_ty, d_tz;       // block id -> tile
  uint32_t ntx, nty, ntz, nblk;
};

template <int NT, int MT, int NP, typename TIN, typename TOUT>
__device__ __forceinline__ void conv_igemm_body(const ConvDev& P) {
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
  // batched passes (group_batch > 0): samples [g*group_batch, (g+1)*group_batch) are one BatchNorm group -- its own affine
  // table on the operand load and its own rows of the statistics accumulator (a tile belongs to one sample)
  const int bgrp = a.group_batch > 0 ? b / a.group_batch : 0;
  const float* const g_in_scale = a.in_scale ? a.in_scale + (size_t)bgrp * a.CPi : nullptr;
  const float* const g_in_shift = a.in_shift ? a.in_shift + (size_t)bgrp * a.CPi : nullptr;
  double* const g_stats = a.stats ? a.stats + (size_t)bgrp * a.stats_nrep * a.CPo * 2 : nullptr;
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
  const TIN* __restrict__ xpl = reinterpret_cast<const TIN*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * 16;      // plane-major: sample b of plane 0
  const bf16x8* __restrict__ wf_hi = reinterpret_cast<const bf16x8*>(a.wfrag_hi);
  const bf16x8* __restrict__ wf_lo = reinterpret_cast<const bf16x8*>(a.wfrag_lo);
  const int nvox_tile = a.ITD * a.ITH * a.ITW;
  const int nchunks = nvox_tile * a.octs_per_group;

  // a thread's channel octet is the same for every chunk it stages when the octets per group divide 256
  const bool fixed_oc = (256 % a.octs_per_group) == 0;
  const int my_oc = tid % a.octs_per_group;

  STAMP(0);
  for (int grp = 0; grp < a.ngroups; ++grp) {
    if (grp > 0) __syncthreads();   // previous group's reads are done
    // ---- stage the halo tile: global (channels-last) -> norm -> bf16 (hi/lo) -> LDS planes ----
    // SB independent 16-byte loads per thread are issued before any of them is consumed: the tile
    // load is latency-bound otherwise (one 1 KiB wave-load in flight per wave).
    const int oct0 = grp * a.octs_per_group;
    float fsc[8], fsh[8];
    if (g_in_scale && fixed_oc) {
      const int c = (oct0 + my_oc) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { fsc[j] = g_in_scale[c + j]; fsh[j] = g_in_shift[c + j]; }
    }
    for (int base = 0; base < nchunks; base += 256 * SB) {
      RawChunk<TIN> raw[SB];
      int dsto[SB], cch[SB];
      unsigned valid = 0, inb = 0;
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        // no branch around the load (a divergent join would force vmcnt(0) per load): clamp the chunk
        // index and the coordinates to something loadable, remember validity in bit masks
        const int i0 = base + u * 256 + tid;
        const int i = i0 < nchunks ? i0 : nchunks - 1;
        const uint32_t vox = fdiv(i, P.d_octs);
        const int oc = i - vox * a.octs_per_group;
        const uint32_t row = fdiv(vox, P.d_itw);
        const int vx = vox - row * a.ITW;
        const uint32_t vz = fdiv(row, P.d_ith);
        const int vy = row - vz * a.ITH;
        const int gz = iz0 + (int)vz, gy = iy0 + vy, gx = ix0 + vx;
        const int pl = oc / a.opp, po = oc - pl * a.opp;
        dsto[u] = pl * a.plane_bytes + vox * a.vsb + po * 16;
        cch[u] = (oct0 + oc) * 8;
        const bool ok = (unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi;
        valid |= (i0 < nchunks ? 1u : 0u) << u;
        inb |= (ok ? 1u : 0u) << u;
        const int cz = min(max(gz, 0), a.Di - 1), cy = min(max(gy, 0), a.Hi - 1), cx = min(max(gx, 0), a.Wi - 1);
        // (x_plane != 0: plane-major input [CPi/16][B][D][H][W][16], the concat buffers of the U-Net's up path)
        const size_t vo_ = ((size_t)cz * a.Hi + cy) * a.Wi + cx;
        raw[u].ld(a.x_plane ? xpl + (size_t)(cch[u] >> 4) * (size_t)a.x_plane + vo_ * 16 + (cch[u] & 15) : xin + vo_ * a.CPi + cch[u], a.x_lo_delta);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        if (valid & (1u << u)) {
          float v[8];
          if (inb & (1u << u)) {
            raw[u].unpack(v);
            if (g_in_scale) {
              if (fixed_oc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], fsc[j], fsh[j]);
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], g_in_scale[cch[u] + j], g_in_shift[cch[u] + j]);
              }
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
          }
          unsigned char* dst = tile + dsto[u];
          uint32_t w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (uint32_t)f2bf(v[2 * j]) | ((uint32_t)f2bf(v[2 * j + 1]) << 16);
          *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
          if (NP == 2) {
            uint32_t wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float r0 = v[2 * j] - sp_h2f_lo(w[j]);
              const float r1 = v[2 * j + 1] - sp_h2f_hi(w[j]);
              wl[j] = (uint32_t)f2bf(r0) | ((uint32_t)f2bf(r1) << 16);
            }
            *reinterpret_cast<uint4*>(dst + a.lo_offset) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
          }
        }
      }
    }

    // ---- K loop: taps x channel octets of this group ------------------------------------------
    // weight fragments (L2/L1-resident, shared by every workgroup) and the ktab entry of step s+1 are
    // fetched while step s computes; the first fetch is issued before the barrier that ends staging.
    const size_t gstep0 = (size_t)grp * a.steps_per_group;
    const size_t fstride = (size_t)a.NTtot * 64;
    const bf16x8* wp_hi = wf_hi + (gstep0 * a.NTtot + nt0) * 64 + lane;
    const bf16x8* wp_lo = wf_lo + (gstep0 * a.NTtot + nt0) * 64 + lane;
    bf16x8 wa_n[NT], wl_n[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      wa_n[n] = wp_hi[(size_t)n * 64];
      if (NP == 2) wl_n[n] = wp_lo[(size_t)n * 64];
    }
    if (grp == 0) STAMP(1);
    __syncthreads();
    if (grp == 0) STAMP(2);
    if constexpr (NP == 1) {
      // bf16: weight fragments fetched THREE steps ahead (an L2 round trip is longer than one step's MFMAs; the FC-like CAE
      // layers stream 750 KB of fragments per workgroup and were latency-bound on them)
      const int nst = a.steps_per_group;
      bf16x8 w1[NT], w2[NT], w3[NT];
#define SP_LDW(dst, step_)                                                                            \
  { const int st_ = (step_) < nst ? (step_) : nst - 1;                                                \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) dst[n] = wp_hi[(size_t)st_ * fstride + (size_t)n * 64]; }
#define SP_STEP(wv, step_)                                                                            \
  { const int koff = ktab_l[(step_) * 4 + lg];                                                        \
    bf16x8 xb[MT];                                                                                    \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) xb[m] = *reinterpret_cast<const bf16x8*>(tile + vbase[m] + koff); \
    _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                    \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[n][m] = SP_MFMA16(wv[n], xb[m], acc[n][m], 0, 0, 0); }
      SP_LDW(w1, 1)
      SP_LDW(w2, 2)
      for (int s = 0; s < nst; s += 4) {
        SP_LDW(w3, s + 3)
        SP_STEP(wa_n, s)
        if (s + 1 < nst) { SP_LDW(wa_n, s + 4) SP_STEP(w1, s + 1) }
        if (s + 2 < nst) { SP_LDW(w1, s + 5) SP_STEP(w2, s + 2) }
        if (s + 3 < nst) { SP_LDW(w2, s + 6) SP_STEP(w3, s + 3) }
      }
#undef SP_LDW
#undef SP_STEP
    } else {
    int koff_n = ktab_l[lg];
    for (int s = 0; s < a.steps_per_group; ++s) {
      bf16x8 wa[NT], wl[NT];
      const int koff = koff_n;
#pragma unroll
      for (int n = 0; n < NT; ++n) { wa[n] = wa_n[n]; if (NP == 2) wl[n] = wl_n[n]; }
      if (s + 1 < a.steps_per_group) {
        koff_n = ktab_l[(s + 1) * 4 + lg];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          wa_n[n] = wp_hi[(s + 1) * fstride + (size_t)n * 64];
          if (NP == 2) wl_n[n] = wp_lo[(s + 1) * fstride + (size_t)n * 64];
        }
      }
      bf16x8 xb[MT], xl[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        xb[m] = *reinterpret_cast<const bf16x8*>(tile + vbase[m] + koff);
        if (NP == 2) xl[m] = *reinterpret_cast<const bf16x8*>(tile + a.lo_offset + vbase[m] + koff);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[n][m] = SP_MFMA16(wa[n], xb[m], acc[n][m], 0, 0, 0);
          if (NP == 2) {
            acc[n][m] = SP_MFMA16(wa[n], xl[m], acc[n][m], 0, 0, 0);
            acc[n][m] = SP_MFMA16(wl[n], xb[m], acc[n][m], 0, 0, 0);
          }
        }
      }
    }
    }
  }

  STAMP(3);
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
        if (sizeof(TOUT) == 2 && !std::is_same<TOUT, sp_hl_t>::value) {   // statistics of what is actually stored (bf16-rounded values)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = bf2f(f2bf(v[j]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[n][j] += v[j]; s2[n][j] += v[j] * v[j]; }
        const size_t off = (((size_t)(oz * a.osD + a.ooD) * a.YH + (oy * a.osH + a.ooH)) * a.YW + (ox * a.osW + a.ooW)) * a.CPo + c0;
        if constexpr (std::is_same<TOUT, sp_hl_t>::value) sp_hl_st4(yout + off, a.y_lo_delta, v);
        else Store<TOUT>::st4(yout + off, v);
      }
    }
  }

  STAMP(4);
  if (a.stats) {
    __syncthreads();                       // tile no longer needed: reuse LDS for the block reduction
    float* red = reinterpret_cast<float*>(lds);      // [4 waves][NT * 32] (ordered sum: sp_cols_sum)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x1 = row16_sum(s1[n][j]), x2 = row16_sum(s2[n][j]);
        if (lv == 0) {
          red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2] = x1;
          red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2 + 1] = x2;
        }
      }
    __syncthreads();
    for (int i = tid; i < NT * 16 * 2; i += 256) {
      const int c = nt0 * 16 + (i >> 1);
      if (c < a.CPo) atomicAdd(&g_stats[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (i & 1)], (double)sp_cols_sum(red, NT * 32, 4, i));
    }
  }
  STAMP(5);
}

template <int NT, int MT, int NP, typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvDev P) {
  conv_igemm_body<NT, MT, NP, TIN, TOUT>(P);
}

// Several sub-convolutions of ONE op in one launch (the parity classes of a stride-2 transposed convolution, Cae3D.py:
// 178-204: eight launches of 10-22 us, most of them a fraction of a wave of workgroups): blockIdx.z selects the class; the
// classes differ in taps, offsets and extents only, so one template instance serves them all.
#define SP_CONV_MULTI_MAX 8
struct ConvDevMulti { ConvDev p[SP_CONV_MULTI_MAX]; };
template <int NT, int MT, int NP, typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void conv_igemm_multi_kernel(const ConvDevMulti M) {
  const ConvDev& P = M.p[blockIdx.z];
  if (blockIdx.x >= P.nblk) return;
  conv_igemm_body<NT, MT, NP, TIN, TOUT>(P);
}

// ------------------------------------------------------------------------------------------------
template <int NT, int MT, int NP, typename TIN, typename TOUT>
static int launch_conv_multi(const ConvDevMulti& M, int lds_bytes, dim3 grid, hipStream_t st) {
  auto kern = conv_igemm_multi_kernel<NT, MT, NP, TIN, TOUT>;
  SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_igemm_multi");
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, M);
  SP_CHECK_LAUNCH("sp_conv3d_igemm_multi");
  return SP_OK;
}
template <int NT, int MT>
static int dispatch_dtype_multi(const ConvDevMulti& M, int lds_bytes, dim3 grid, hipStream_t st) {
  const int di = M.p[0].a.dtype_in, dout = M.p[0].a.dtype_out;
  if (di == SP_BF16 && dout == SP_BF16) return launch_conv_multi<NT, MT, 1, bf16_t, bf16_t>(M, lds_bytes, grid, st);
  if (di == SP_F32 && dout == SP_F32) return launch_conv_multi<NT, MT, 2, float, float>(M, lds_bytes, grid, st);
  if (di == SP_BF16 && dout == SP_F32) return launch_conv_multi<NT, MT, 1, bf16_t, float>(M, lds_bytes, grid, st);
  if (di == SP_HL && dout == SP_HL) return launch_conv_multi<NT, MT, 2, sp_hl_t, sp_hl_t>(M, lds_bytes, grid, st);
  sp_set_error("sp_conv3d_igemm_multi: unsupported dtype pair in=%d out=%d", di, dout);
  return SP_EINVAL;
}

template <int NT, int MT, int NP, typename TIN, typename TOUT>
static int launch_conv(const ConvDev& P, dim3 grid, hipStream_t st) {
  auto kern = conv_igemm_kernel<NT, MT, NP, TIN, TOUT>;
  SP_ENSURE_LDS(kern, P.a.lds_bytes, "sp_conv3d_igemm");
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
  if (di == SP_HL && dout == SP_HL) return launch_conv<NT, MT, 2, sp_hl_t, sp_hl_t>(P, grid, st);      // bf16 pairs: hi / lo go to LDS as they are
  sp_set_error("sp_conv3d_igemm: unsupported dtype pair in=%d out=%d", di, dout);
  return SP_EINVAL;
}

int sp_conv3d_igemm_dma(const sp_conv_args* a, sp_stream_t stream);   // sp_conv_dma.hip

static int conv_check_build(const sp_conv_args* a, ConvDev& P);

// n sub-convolutions (n <= 8) that share weights layout, register blocking and data types -- the parity classes of a
// transposed / strided-gradient op -- in ONE launch of the register-staged kernel; anything else (DMA-staged classes,
// differing blockings) falls back to one sp_conv3d_igemm per class.
extern "C" int sp_conv3d_igemm_multi(const sp_conv_args* args, int32_t n, sp_stream_t stream) {
  SP_CHECK_ARG(args && n >= 1, "sp_conv3d_igemm_multi: bad arguments");
  bool same = n >= 2 && n <= SP_CONV_MULTI_MAX;
  for (int i = 0; i < n && same; ++i)
    same = !args[i].dma && args[i].NT == args[0].NT && args[i].MT == args[0].MT && args[i].NTtot == args[0].NTtot &&
           args[i].dtype_in == args[0].dtype_in && args[i].dtype_out == args[0].dtype_out;
  if (!same) {
    for (int i = 0; i < n; ++i) {
      const int rc = sp_conv3d_igemm(&args[i], stream);
      if (rc != SP_OK) return rc;
    }
    return SP_OK;
  }
  ConvDevMulti M;
  int lds = 0;
  uint32_t gx = 0;
  for (int i = 0; i < n; ++i) {
    const int rc = conv_check_build(&args[i], M.p[i]);
    if (rc != SP_OK) return rc;
    lds = args[i].lds_bytes > lds ? args[i].lds_bytes : lds;
    gx = M.p[i].nblk > gx ? M.p[i].nblk : gx;
  }
  for (int i = n; i < SP_CONV_MULTI_MAX; ++i) M.p[i] = M.p[0];
  dim3 grid(gx, args[0].NTtot / args[0].NT, n);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const sp_conv_args* a = &args[0];
#define SP_CASE(NT_, MT_) if (a->NT == NT_ && a->MT == MT_) return dispatch_dtype_multi<NT_, MT_>(M, lds, grid, st)
  SP_CASE(1, 8); SP_CASE(2, 8); SP_CASE(3, 8); SP_CASE(4, 8);
  SP_CASE(1, 4); SP_CASE(2, 4); SP_CASE(3, 4); SP_CASE(4, 4);
  SP_CASE(1, 2); SP_CASE(2, 2); SP_CASE(3, 2); SP_CASE(4, 2);
#undef SP_CASE
  sp_set_error("sp_conv3d_igemm_multi: no kernel for NT=%d MT=%d", a->NT, a->MT);
  return SP_EINVAL;
}

static int conv_check_build(const sp_conv_args* a, ConvDev& P) {
  SP_CHECK_ARG(a && a->x && a->y && a->wfrag_hi && a->ktab, "sp_conv3d_igemm: null pointer");
  SP_CHECK_ARG(a->CPi % 8 == 0 && a->CPo % 8 == 0, "sp_conv3d_igemm: channel pitch must be a multiple of 8 (CPi=%d CPo=%d)", a->CPi, a->CPo);
  SP_CHECK_ARG(a->TD * a->TH == 4 * a->MT, "sp_conv3d_igemm: TD*TH (%d*%d) must equal 4*MT (%d)", a->TD, a->TH, 4 * a->MT);
  SP_CHECK_ARG(a->NTtot % a->NT == 0, "sp_conv3d_igemm: NTtot %d not a multiple of NT %d", a->NTtot, a->NT);
  SP_CHECK_ARG(a->ngroups * a->octs_per_group * 8 == a->CPi, "sp_conv3d_igemm: groups (%d x %d octets) do not cover CPi=%d", a->ngroups, a->octs_per_group, a->CPi);
  SP_CHECK_ARG(a->dtype_in == SP_BF16 || (a->wfrag_lo && a->lo_offset > 0), "sp_conv3d_igemm: f32 / bf16-pair mode needs wfrag_lo and lo_offset");
  SP_CHECK_ARG(a->dtype_in != SP_HL || (a->x_lo_delta != 0 && a->x_lo_delta % 16 == 0), "sp_conv3d_igemm: bf16 pair input needs x_lo_delta");
  SP_CHECK_ARG(a->dtype_out != SP_HL || (a->y_lo_delta != 0 && a->y_lo_delta % 8 == 0), "sp_conv3d_igemm: bf16 pair output needs y_lo_delta");
  SP_CHECK_ARG(a->Do > 0 && a->Ho > 0 && a->Wo > 0 && a->B > 0, "sp_conv3d_igemm: empty output");
  SP_CHECK_ARG(!a->stats || (a->stats_nrep >= 1 && (a->stats_nrep & (a->stats_nrep - 1)) == 0), "sp_conv3d_igemm: stats_nrep must be a power of two");
  // the staged tile must cover every tap of every output row of the tile
  SP_CHECK_ARG(a->ITW >= 15 * a->sW + 1 && a->ITH >= (a->TH - 1) * a->sH + 1 && a->ITD >= (a->TD - 1) * a->sD + 1, "sp_conv3d_igemm: input tile smaller than output tile");
  {
    const int planes = (a->octs_per_group + a->opp - 1) / a->opp;
    const long tile_bytes = (long)planes * a->plane_bytes * (a->dtype_in != SP_BF16 ? 2 : 1);
    const long need = ((a->steps_per_group * 16 + 15) & ~15) + tile_bytes;
    SP_CHECK_ARG(a->plane_bytes >= a->ITD * a->ITH * a->ITW * a->vsb && need <= a->lds_bytes && a->lds_bytes <= 160 * 1024,
                 "sp_conv3d_igemm: LDS plan inconsistent (need %ld, lds_bytes %d)", need, a->lds_bytes);
    SP_CHECK_ARG(a->lds_bytes >= a->NT * 16 * 2 * 4, "sp_conv3d_igemm: LDS too small for the reduction");
  }
  SP_CHECK_ARG(a->stats_mode == 0 || (a->dma && a->aux), "sp_conv3d_igemm: stats_mode 1 needs the DMA kernel and aux");
  SP_CHECK_ARG(a->x_plane == 0 || a->dma || (a->CPi % 16 == 0 && a->x_plane >= (int64_t)a->B * a->Di * a->Hi * a->Wi * 16),
               "sp_conv3d_igemm: plane-major input needs whole 16-channel planes (x_plane >= one plane)");
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
  return SP_OK;
}

extern "C" int sp_conv3d_igemm(const sp_conv_args* a, sp_stream_t stream) {
  ConvDev P;
  const int rc = conv_check_build(a, P);
  if (rc != SP_OK) return rc;
  SP_CHECK_ARG(a->group_batch == 0 || (a->group_batch > 0 && a->B % a->group_batch == 0), "sp_conv3d_igemm: group_batch %d does not divide the batch %d", a->group_batch, a->B);
  if (a->dma) return sp_conv3d_igemm_dma(a, stream);
  dim3 grid(P.nblk, a->NTtot / a->NT);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define SP_CASE(NT_, MT_) if (a->NT == NT_ && a->MT == MT_) return dispatch_dtype<NT_, MT_>(P, grid, st)
  SP_CASE(1, 8); SP_CASE(2, 8); SP_CASE(3, 8); SP_CASE(4, 8);
  SP_CASE(1, 4); SP_CASE(2, 4); SP_CASE(3, 4); SP_CASE(4, 4);
  SP_CASE(1, 2); SP_CASE(2, 2); SP_CASE(3, 2); SP_CASE(4, 2);
#undef SP_CASE
  sp_set_error("sp_conv3d_igemm: no kernel for NT=%d MT=%d", a->NT, a->MT);
  return SP_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// weight re-packing: fp32 (Cout,Cin,taps) -> MFMA A fragments [step][ntile][lane][8] (hi / lo bf16)
__device__ __forceinline__ void prep_wfrag_one(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin,
                                               const int32_t* __restrict__ kmap, int nsteps, int NTtot,
                                               bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                               const float* __restrict__ fold, int64_t idx) {   // idx = (step, ntile, lane)
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
        if (co < Cout && ci < Cin) v = w[co * sCo + ci * sCi + tap] * (fold ? fold[ci] : 1.f);
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
__global__ void prep_wfrag_kernel(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin,
                                  const int32_t* __restrict__ kmap, int nsteps, int NTtot,
                                  bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, const float* __restrict__ fold) {
  prep_wfrag_one(w, sCo, sCi, Cout, Cin, kmap, nsteps, NTtot, hi, lo, fold, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// all re-packs that depend on the parameters only (the data-gradient weights of every layer) in ONE launch: grid.y = item
__global__ void prep_wfrag_batch_kernel(const sp_prep_item* __restrict__ items) {
  const sp_prep_item it = items[blockIdx.y];
  if (blockIdx.x == 0 && it.bias_out)      // the layer's bias, zero-padded to whole output tiles
    for (int c = threadIdx.x; c < it.bias_pad; c += blockDim.x) it.bias_out[c] = (it.bias && c < it.bias_n) ? it.bias[c] : 0.f;
  prep_wfrag_one(it.w, it.sCo, it.sCi, it.Cout, it.Cin, it.kmap, it.nsteps, it.NTtot, reinterpret_cast<bf16_t*>(it.wfrag_hi),
                 reinterpret_cast<bf16_t*>(it.wfrag_lo), it.fold_scale, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
extern "C" int sp_conv_prep_weights_batch(const sp_prep_item* items_dev, int32_t n, int32_t max_blocks, sp_stream_t stream) {
  SP_CHECK_ARG(items_dev && n >= 1 && n <= 65535 && max_blocks >= 1, "sp_conv_prep_weights_batch: bad arguments");
  hipLaunchKernelGGL(prep_wfrag_batch_kernel, dim3((unsigned)max_blocks, (unsigned)n), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), items_dev);
  SP_CHECK_LAUNCH("sp_conv_prep_weights_batch");
  return SP_OK;
}

extern "C" int sp_conv_prep_weights(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin,
                                    const int32_t* kmap, int32_t nsteps, int32_t NTtot, void* wfrag_hi,
                                    void* wfrag_lo, const float* fold_scale, sp_stream_t stream) {
  SP_CHECK_ARG(w && kmap && wfrag_hi && nsteps > 0 && NTtot > 0, "sp_conv_prep_weights: bad arguments");
  const int64_t total = (int64_t)nsteps * NTtot * 64;
  hipLaunchKernelGGL(prep_wfrag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, sCo, sCi, Cout, Cin, kmap, nsteps, NTtot,
                     reinterpret_cast<bf16_t*>(wfrag_hi), reinterpret_cast<bf16_t*>(wfrag_lo), fold_scale);
  SP_CHECK_LAUNCH("sp_conv_prep_weights");
  return SP_OK;
}

__global__ void fold_bias_kernel(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin, int ntaps,
                                 const float* __restrict__ bias, const float* __restrict__ shift,
                                 float* __restrict__ out, int CoutPad) {
  const int co = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  if (co < Cout)
    for (int i = lane; i < Cin * ntaps; i += 64) {
      const int ci = i / ntaps, tp = i - ci * ntaps;
      s += w[co * sCo + ci * sCi + tp] * shift[ci];
    }
  s = wave_sum(s);
  if (lane == 0) out[co] = co < Cout ? s + (bias ? bias[co] : 0.f) : 0.f;
}
// re-pack + BatchNorm fold of the weights AND of the bias in one launch (the two kernels above, blocks [0, nprep)
// and [nprep, nprep + CoutPad/4)): one ~5 us dispatch less per layer and step
// BN: the BatchNorm finalize runs inside (sp_conv_prep_folded_bn): every workgroup derives scale / shift of the Cin input channels
// into LDS (sp_bn_fin_block: the arithmetic of sp_bn_finalize), workgroup 0 publishes them and the running statistics
template <bool BN>
__global__ __launch_bounds__(256) void prep_folded_kernel(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin,
                                                           const int32_t* __restrict__ kmap, int nsteps, int NTtot,
                                                           bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                           const float* __restrict__ fold, int nprep, int ntaps,
                                                           const float* __restrict__ bias, const float* __restrict__ shift,
                                                           float* __restrict__ out, int CoutPad, const sp_bn_fin_args bn) {
  extern __shared__ float bn_lds[];      // BN: [2][bn.CP] scale, shift
  if constexpr (BN) {
    sp_bn_fin_block(bn, blockIdx.x == 0, bn_lds, bn_lds + bn.CP);
    fold = bn_lds;
    shift = bn_lds + bn.CP;
  }
  if ((int)blockIdx.x < nprep) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (step, ntile, lane)
    const int64_t total = (int64_t)nsteps * NTtot * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int nt = (idx >> 6) % NTtot;
    const int st = (idx >> 6) / NTtot;
    const int co = nt * 16 + (lane & 15), g = lane >> 4;
    const int km = kmap[st * 4 + g];
    uint32_t wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v = 0.f;
        if (km >= 0) {
          const int tap = km >> 16, ci = (km & 0xffff) * 8 + 2 * j + h;
          if (co < Cout && ci < Cin) v = w[co * sCo + ci * sCi + tap] * fold[ci];
        }
        f[h] = v;
      }
      const bf16_t h0 = f2bf(f[0]), h1 = f2bf(f[1]);
      wh[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      wl[j] = (uint32_t)f2bf(f[0] - bf2f(h0)) | ((uint32_t)f2bf(f[1] - bf2f(h1)) << 16);
    }
    reinterpret_cast<uint4*>(hi)[idx] = make_uint4(wh[0], wh[1], wh[2], wh[3]);
    if (lo) reinterpret_cast<uint4*>(lo)[idx] = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    return;
  }
  // folded bias of output channel co: one workgroup per channel, eight independent loads per thread and trip (one wave walking
  // Cin * ntaps = 1296 ... 2592 products in a plain loop paid 20-40 dependent round trips: 6-12 us of a 7-23 us launch)
  __shared__ float bred[4];
  const int co = (int)blockIdx.x - nprep;
  if (co >= CoutPad) return;
  float acc = 0.f;
  if (co < Cout) {
    const int n = Cin * ntaps;
    const float* wr = w + co * sCo;
    int i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {
      float wv[8], sv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = i + 256 * k, ci = e / ntaps, tp = e - ci * ntaps;
        wv[k] = wr[ci * sCi + tp];
        sv[k] = shift[ci];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fmaf(wv[k], sv[k], acc);
    }
    for (; i < n; i += 256) {
      const int ci = i / ntaps, tp = i - ci * ntaps;
      acc = fmaf(wr[ci * sCi + tp], shift[ci], acc);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) bred[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[co] = co < Cout ? ((bred[0] + bred[1]) + (bred[2] + bred[3])) + (bias ? bias[co] : 0.f) : 0.f;
}
extern "C" int sp_conv_prep_folded(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap,
                                   int32_t nsteps, int32_t NTtot, void* wfrag_hi, void* wfrag_lo, const float* fold_scale,
                                   int32_t ntaps, const float* bias, const float* fold_shift, float* bias_out,
                                   int32_t CoutPad, sp_stream_t stream) {
  SP_CHECK_ARG(w && kmap && wfrag_hi && fold_scale && fold_shift && bias_out && nsteps > 0 && NTtot > 0 && CoutPad >= Cout,
               "sp_conv_prep_folded: bad arguments");
  const int64_t total = (int64_t)nsteps * NTtot * 64;
  const int nprep = (int)((total + 255) / 256);
  hipLaunchKernelGGL(prep_folded_kernel<false>, dim3((unsigned)(nprep + CoutPad)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, sCo, sCi, Cout, Cin, kmap, nsteps, NTtot,
                     reinterpret_cast<bf16_t*>(wfrag_hi), reinterpret_cast<bf16_t*>(wfrag_lo), fold_scale, nprep, ntaps, bias,
                     fold_shift, bias_out, CoutPad, sp_bn_fin_args{});
  SP_CHECK_LAUNCH("sp_conv_prep_folded");
  return SP_OK;
}
extern "C" int sp_conv_prep_folded_bn(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap,
                                      int32_t nsteps, int32_t NTtot, void* wfrag_hi, void* wfrag_lo, int32_t ntaps, const float* bias,
                                      float* bias_out, int32_t CoutPad, const sp_bn_fin_args* bn, sp_stream_t stream) {
  SP_CHECK_ARG(w && kmap && wfrag_hi && bias_out && bn && nsteps > 0 && NTtot > 0 && CoutPad >= Cout, "sp_conv_prep_folded_bn: bad arguments");
  SP_CHECK_ARG(bn->gamma && bn->beta && bn->scale && bn->shift && bn->C == Cin && bn->CP >= Cin && bn->CP <= 4096 && bn->nrep >= 1 &&
               (bn->training ? (bn->sums != nullptr && bn->count > 0) : (bn->running_mean && bn->running_var)),
               "sp_conv_prep_folded_bn: BatchNorm arguments (C %d for %d input channels, pitch %d, %s statistics)", bn->C, Cin, bn->CP,
               bn->training ? "batch" : "running");
  const int64_t total = (int64_t)nsteps * NTtot * 64;
  const int nprep = (int)((total + 255) / 256);
  hipLaunchKernelGGL(prep_folded_kernel<true>, dim3((unsigned)(nprep + CoutPad)), dim3(256), (size_t)bn->CP * 2 * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), w, sCo, sCi, Cout, Cin, kmap, nsteps, NTtot,
                     reinterpret_cast<bf16_t*>(wfrag_hi), reinterpret_cast<bf16_t*>(wfrag_lo), (const float*)nullptr, nprep, ntaps, bias,
                     (const float*)nullptr, bias_out, CoutPad, *bn);
  SP_CHECK_LAUNCH("sp_conv_prep_folded_bn");
  return SP_OK;
}

// ---- BatchNorm folded per GROUP into a padded stride-1 3x3x3 convolution (the CAE's batched passes, sp_conv_args.bias_tab):
// blocks [0, G * nprep): the fragments of group g with fold = its scale row; the rest: one wave per table entry
// (g, class, co) = b[co] + sum over the taps that are valid for the class of sum_ci W[co][ci][tap] * shift_g[ci]
__device__ __forceinline__ bool tap_valid_for_class(int t, int c, int p) {      // axis with padding p (k = 3): class c of 2 p + 1
  return c < p ? t >= p - c : (c == p ? true : t < 3 - (c - p));
}
__global__ __launch_bounds__(256) void prep_folded_groups_kernel(const float* __restrict__ w, int64_t sCo, int64_t sCi, int Cout, int Cin,
                                                                  const int32_t* __restrict__ kmap, int nsteps, int NTtot,
                                                                  unsigned char* __restrict__ wfrag, int64_t frag_gstride,
                                                                  const float* __restrict__ coef, int coef_gstride, int coef_pitch, int G,
                                                                  int nprep, const float* __restrict__ bias, int pD, int pH, int pW,
                                                                  float* __restrict__ tab, int CoutPad) {
  if ((int)blockIdx.x < G * nprep) {
    const int g = blockIdx.x / nprep, blk = blockIdx.x - g * nprep;
    prep_wfrag_one(w, sCo, sCi, Cout, Cin, kmap, nsteps, NTtot, reinterpret_cast<bf16_t*>(wfrag + (size_t)g * frag_gstride), nullptr,
                   coef + (size_t)g * coef_gstride, (int64_t)blk * 256 + threadIdx.x);
    return;
  }
  const int nz = 2 * pD + 1, ny = 2 * pH + 1, nx = 2 * pW + 1, ncls = nz * ny * nx;
  const int e = ((int)blockIdx.x - G * nprep) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= G * ncls * CoutPad) return;
  const int co = e % CoutPad, cls = (e / CoutPad) % ncls, g = e / (CoutPad * ncls);
  const int cx = cls % nx, cy = (cls / nx) % ny, cz = cls / (nx * ny);
  const float* shift = coef + (size_t)g * coef_gstride + 2 * coef_pitch;
  float acc = 0.f;
  if (co < Cout)
    for (int i = lane; i < Cin * 27; i += 64) {
      const int ci = i / 27, tp = i - ci * 27;
      const int tz = tp / 9, ty = (tp / 3) % 3, tx = tp % 3;
      if (tap_valid_for_class(tz, cz, pD) && tap_valid_for_class(ty, cy, pH) && tap_valid_for_class(tx, cx, pW))
        acc += w[co * sCo + ci * sCi + tp] * shift[ci];
    }
  acc = wave_sum(acc);
  if (lane == 0) tab[e] = co < Cout ? acc + (bias ? bias[co] : 0.f) : 0.f;
}
extern "C" int sp_conv_prep_folded_groups(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap, int32_t nsteps,
                                          int32_t NTtot, void* wfrag, int64_t frag_gstride, const float* coef, int32_t coef_gstride,
                                          int32_t coef_pitch, int32_t G, const float* bias, int32_t padD, int32_t padH, int32_t padW,
                                          float* bias_tab, int32_t CoutPad, sp_stream_t stream) {
  SP_CHECK_ARG(w && kmap && wfrag && coef && bias_tab && nsteps > 0 && NTtot > 0 && G >= 1 && CoutPad >= Cout && coef_pitch >= Cin && coef_gstride >= 3 * coef_pitch,
               "sp_conv_prep_folded_groups: bad arguments");
  SP_CHECK_ARG(padD >= 0 && padD <= 2 && padH >= 0 && padH <= 2 && padW >= 0 && padW <= 2, "sp_conv_prep_folded_groups: padding 0..2");
  const int64_t total = (int64_t)nsteps * NTtot * 64;
  SP_CHECK_ARG(frag_gstride >= total * 16 && frag_gstride % 16 == 0, "sp_conv_prep_folded_groups: frag_gstride %lld < one fragment set (%lld bytes)", (long long)frag_gstride, (long long)total * 16);
  const int nprep = (int)((total + 255) / 256);
  const int ncls = (2 * padD + 1) * (2 * padH + 1) * (2 * padW + 1);
  const int ntab = (G * ncls * CoutPad + 3) / 4;
  hipLaunchKernelGGL(prep_folded_groups_kernel, dim3((unsigned)(G * nprep + ntab)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, sCo, sCi,
                     Cout, Cin, kmap, nsteps, NTtot, reinterpret_cast<unsigned char*>(wfrag), frag_gstride, coef, coef_gstride, coef_pitch, G, nprep,
                     bias, padD, padH, padW, bias_tab, CoutPad);
  SP_CHECK_LAUNCH("sp_conv_prep_folded_groups");
  return SP_OK;
}

extern "C" int sp_conv_fold_bias(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, int32_t ntaps,
                                 const float* bias, const float* shift, float* bias_out, int32_t CoutPad,
                                 sp_stream_t stream) {
  SP_CHECK_ARG(w && shift && bias_out && CoutPad >= Cout, "sp_conv_fold_bias: bad arguments");
  hipLaunchKernelGGL(fold_bias_kernel, dim3(CoutPad), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), w, sCo, sCi,
                     Cout, Cin, ntaps, bias, shift, bias_out, CoutPad);
  SP_CHECK_LAUNCH("sp_conv_fold_bias");
  return SP_OK;
}
