// bf16 fast path of the implicit-GEMM convolution (see sp_conv.hip for the GEMM view and the tables).
//
// What the phase stamps of the register-staged kernel showed on the 16->16 @126^3 layer (29 K cycles per
// workgroup: 51 % staging, 24 % K loop, 17 % epilogue, 7 % statistics) drives this variant:
//   * the halo tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction,
//     no VGPR round trip, no conversion): all of a workgroup's tile is in flight at once.  The plane
//     layout [plane][voxel][opp x 16 B] is lane-linear, so chunk i of a group lands at byte 16*i;
//   * BatchNorm cannot be applied on a DMA: the host folds it into the weights and the bias
//     (conv(s*x+t) = conv_{W*s}(x) + sum W*t, exact for un-padded convolutions) -- in_scale must be NULL;
//   * zero padding (data gradients, padded convolutions without a norm): a second pass overwrites the
//     out-of-volume chunks with zeros, only in workgroups whose tile crosses the volume border;
//   * layers whose K loop has <= 16 weight fragments per group keep them ALL in registers (loaded while
//     the DMA is in flight); longer loops prefetch one step ahead.
#include "sp_common.h"

struct ConvDmaDev {
  sp_conv_args a;
  FastDiv d_itw, d_ith, d_plane;  // staging index math
  FastDiv d_tx, d_ty, d_tz;       // block id -> tile
  uint32_t ntx, nty, ntz, nblk;
  int32_t plane_chunks, group_chunks, nruns, log2_opp;
  int32_t row_chunks, segs_per_row, nrows, njobs, buf_stride;
  FastDiv d_segs, d_rows;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;


#ifdef SP_CONV_STAMPS
#define SP_NSTAMP 6
extern __device__ unsigned long long sp_stamp_buf[32768][SP_NSTAMP];
#define STAMP(k)                                                                              \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    unsigned long long t_;                                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 32768) sp_stamp_buf[blockIdx.x][k] = t_; \
  } while (0)
#else
#define STAMP(k)
#endif

// KS > 0: compile-time number of K steps per group, all weight fragments of the group resident in registers
// (no guards inside the unrolled loop: guards make hipcc shuttle the accumulators between VGPRs and AGPRs
// around every step).  KS == 0: run-time step count (even, the planner pads), fragments prefetched one step ahead.
// HL (bf16 pairs, the forward of the "bf16x3" / "f16x3" modes; run-time K loop only): the hi and the lo tile are staged side by
// side (a.lo_offset behind each other, the lo halves of x a.x_lo_delta bytes behind the hi ones), hi and lo weight fragments
// stream together, a product is three MFMAs (w_hi x_hi + w_hi x_lo + w_lo x_hi) into one accumulator, the output is split again.
template <int NT, int MT, int KS, typename TOUT, bool HL = false>
__global__ __launch_bounds__(256, ((NT == 1 && KS > 0) ? 4 : 1)) void conv_igemm_dma_kernel(const ConvDmaDev P) {
  static_assert(!HL || KS == 0, "bf16 pairs: run-time K loop only");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_conv_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lv = lane & 15, lg = lane >> 4;

  uint32_t t = xcd_remap(blockIdx.x, P.nblk);
  uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
  q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; t = q;
  q = fdiv(t, P.d_tz); const int tz = t - q * P.ntz; const int b = q;
  const int oz0 = tz * a.TD, oy0 = ty * a.TH, ox0 = tx * 16;
  const int nt0 = blockIdx.y * NT;
  const int iz0 = oz0 * a.sD + a.o0D, iy0 = oy0 * a.sH + a.o0H, ix0 = ox0 * a.sW + a.o0W;

  int* ktab_l = reinterpret_cast<int*>(lds);
  const int ktab_bytes = (a.steps_per_group * 16 + 15) & ~15;
  unsigned char* tile = lds + ktab_bytes;
  for (int i = tid; i < a.steps_per_group * 4; i += 256) ktab_l[i] = a.ktab[i];

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

  // x_plane != 0: plane-major input [plane][B][D][H][W][16] (concat buffers: each 16-channel plane dense, so a
  // single-plane pass fetches only its own bytes instead of 32-byte slices of 96-byte rows)
  const int xpitch = a.x_plane ? 16 : a.CPi;
  const bf16_t* __restrict__ xin = reinterpret_cast<const bf16_t*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * xpitch;
  const bf16x8* __restrict__ wf_hi = reinterpret_cast<const bf16x8*>(a.wfrag_hi);
  const bool border = a.zfill && (iz0 < 0 || iy0 < 0 || ix0 < 0 || iz0 + a.ITD > a.Di || iy0 + a.ITH > a.Hi || ix0 + a.ITW > a.Wi);
  const int opp_mask = a.opp - 1;

  // ---- DMA job table: job = (plane, tile row, 64-chunk segment); a tile row (ITW voxels x opp octets) is
  // contiguous in global memory and in its LDS plane.  Lane L decodes job L (+64k) ONCE; the issue loop then
  // only broadcasts (v_readlane) the row's element offset and LDS offset: the scalar unit is shared by every
  // wave of the CU, per-job scalar index arithmetic serialises there.
  int job_goff[2], job_loff[2];          // up to 128 jobs per group (host-checked)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int j = k * 64 + lane;
    j = j < P.njobs ? j : P.njobs - 1;
    const uint32_t prow = fdiv(j, P.d_segs);
    const int seg = j - prow * P.segs_per_row;
    const uint32_t pl = fdiv(prow, P.d_rows);
    const int row = prow - pl * P.nrows;
    const uint32_t vz = fdiv(row, P.d_ith);
    const int vy = row - vz * a.ITH;
    const int cz = min(max(iz0 + (int)vz, 0), a.Di - 1), cy = min(max(iy0 + vy, 0), a.Hi - 1);
    job_goff[k] = ((cz * a.Hi + cy) * a.Wi) * xpitch + (a.x_plane ? (int)pl * (int)a.x_plane : (int)pl * a.opp * 8);   // elements; the x / octet part is per lane
    job_loff[k] = (int)pl * a.plane_bytes + row * P.row_chunks * 16 + seg * 1024 + (seg << 24);   // seg kept in the top byte
  }
  STAMP(0);
  for (int grp = 0; grp < a.ngroups; ++grp) {
    if (grp > 0) __syncthreads();
    const int oct0 = grp * a.octs_per_group;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int nj = min(64, P.njobs - k * 64);
      for (int j = wave; j < nj; j += 4) {
        const int goff = __builtin_amdgcn_readlane(job_goff[k], j);
        const int lo = __builtin_amdgcn_readlane(job_loff[k], j);
        const int seg = lo >> 24;
        const int ch = seg * 64 + lane;
        if (ch < P.row_chunks) {
          const int cx = min(max(ix0 + (ch >> P.log2_opp), 0), a.Wi - 1);
          const bf16_t* src = a.x_plane ? xin + goff + (size_t)(oct0 >> 1) * a.x_plane + (cx * 16 + (ch & opp_mask) * 8)
                                        : xin + goff + (cx * a.CPi + (oct0 + (ch & opp_mask)) * 8);
#ifndef SP_NO_DMA
          sp_dma16(src, tile + (lo & 0xffffff));
          if constexpr (HL) sp_dma16(reinterpret_cast<const unsigned char*>(src) + a.x_lo_delta, tile + a.lo_offset + (lo & 0xffffff));
#endif
        }
      }
    }
    // ---- weight fragments of this group: resident (KS > 0) or first step of the prefetch chain ------------
    const size_t gstep0 = (size_t)grp * a.steps_per_group;
    const size_t fstride = (size_t)a.NTtot * 64;
    const bf16x8* wp = wf_hi + (gstep0 * a.NTtot + nt0) * 64 + lane;
    const bf16x8* wpl = HL ? reinterpret_cast<const bf16x8*>(a.wfrag_lo) + (gstep0 * a.NTtot + nt0) * 64 + lane : nullptr;
    bf16x8 wreg[KS > 0 ? KS : 1][NT];
    bf16x8 wa0[NT], wa1[NT];
    bf16x8 wl0[HL ? NT : 1], wl1[HL ? NT : 1], wl2[HL ? NT : 1], wl3[HL ? NT : 1];
    if (KS > 0) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) wreg[s][n] = wp[s * fstride + (size_t)n * 64];
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) wa0[n] = wp[(size_t)n * 64];
      if constexpr (HL) {
#pragma unroll
        for (int n = 0; n < NT; ++n) wl0[n] = wpl[(size_t)n * 64];
      }
    }
    if (grp == 0) STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grp == 0) STAMP(2);
    if (border) {
      // zero the chunks that lie outside the input volume (padding); LDS byte of chunk i is 16*i
      for (int r = wave; r < P.nruns; r += 4) {
        const int i = r * 64 + lane;
        if (i < P.group_chunks) {
          const uint32_t pl = fdiv(i, P.d_plane);
          const int ii = i - pl * P.plane_chunks;
          const uint32_t vox = (uint32_t)ii >> P.log2_opp;
          const uint32_t row = fdiv(vox, P.d_itw);
          const int vx = vox - row * a.ITW;
          const uint32_t vz = fdiv(row, P.d_ith);
          const int vy = row - vz * a.ITH;
          const int gz = iz0 + (int)vz, gy = iy0 + vy, gx = ix0 + vx;
          if (!((unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi)) {
            *reinterpret_cast<uint4*>(tile + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
            if constexpr (HL) *reinterpret_cast<uint4*>(tile + a.lo_offset + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }

    // ---- K loop: the activation fragments of step s+1 are read from LDS while step s runs on the MFMA pipe;
    // static ping-pong buffers (no register copies)
    bf16x8 x0[MT], x1[MT];
    bf16x8 xl0[HL ? MT : 1], xl1[HL ? MT : 1];
#define SP_LDX(dst, koff_)                                                                            \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(tile + vbase[m] + (koff_));
#define SP_LDXL(dst, koff_)                                                                           \
  if constexpr (HL) { _Pragma("unroll") for (int m = 0; m < MT; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(tile + a.lo_offset + vbase[m] + (koff_)); }
#define SP_MMA(wv, xv)                                                                                \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                      \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[n][m] = SP_MFMA16(wv[n], xv[m], acc[n][m], 0, 0, 0);
    // the two cross terms of a pair product (the hi x hi term is SP_MMA)
#define SP_MMAL(wv, wlv, xv, xlv)                                                                     \
  if constexpr (HL) {                                                                                 \
    _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                    \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                              \
          acc[n][m] = SP_MFMA16(wv[n], xlv[m], acc[n][m], 0, 0, 0);                                   \
          acc[n][m] = SP_MFMA16(wlv[n], xv[m], acc[n][m], 0, 0, 0);                                   \
        }                                                                                             \
  }
    {
      const int k0 = ktab_l[lg];
      SP_LDX(x0, k0)
      SP_LDXL(xl0, k0)
    }
    if (KS > 0) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
          const int kn = ktab_l[(s + 1) * 4 + lg];
          if ((s & 1) == 0) { SP_LDX(x1, kn) } else { SP_LDX(x0, kn) }
        }
        if ((s & 1) == 0) { SP_MMA(wreg[s], x0) } else { SP_MMA(wreg[s], x1) }
      }
    } else {
      // Weight fragments stream from global memory (L2): one step of MFMAs (NT*MT x 16 cycles = 256 cycles for two output
      // tiles) does not cover an L2 round trip (500+ cycles), so the fragments are fetched THREE steps ahead (four register
      // sets); the activation fragments come from LDS one step ahead as before.  steps_per_group is even (host plan).
      const int nst = a.steps_per_group;
      bf16x8 wa2[NT], wa3[NT];
#define SP_LDW(dst, dstl, step_)                                                                      \
  { const int st_ = (step_) < nst ? (step_) : nst - 1;                                                \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) dst[n] = wp[(size_t)st_ * fstride + (size_t)n * 64]; \
    if constexpr (HL) { _Pragma("unroll") for (int n = 0; n < NT; ++n) dstl[n] = wpl[(size_t)st_ * fstride + (size_t)n * 64]; } }
      SP_LDW(wa1, wl1, 1)
      SP_LDW(wa2, wl2, 2)
      for (int s = 0; s < nst; s += 4) {
        SP_LDW(wa3, wl3, s + 3)
        { const int kn = ktab_l[(s + 1) * 4 + lg]; SP_LDX(x1, kn) SP_LDXL(xl1, kn) }
        SP_MMA(wa0, x0)
        SP_MMAL(wa0, wl0, x0, xl0)
        SP_LDW(wa0, wl0, s + 4)
        if (s + 2 < nst) { const int kn = ktab_l[(s + 2) * 4 + lg]; SP_LDX(x0, kn) SP_LDXL(xl0, kn) }
        SP_MMA(wa1, x1)
        SP_MMAL(wa1, wl1, x1, xl1)
        if (s + 2 < nst) {
          SP_LDW(wa1, wl1, s + 5)
          { const int kn = ktab_l[(s + 3) * 4 + lg]; SP_LDX(x1, kn) SP_LDXL(xl1, kn) }
          SP_MMA(wa2, x0)
          SP_MMAL(wa2, wl2, x0, xl0)
          SP_LDW(wa2, wl2, s + 6)
          if (s + 4 < nst) { const int kn = ktab_l[(s + 4) * 4 + lg]; SP_LDX(x0, kn) SP_LDXL(xl0, kn) }
          SP_MMA(wa3, x1)
          SP_MMAL(wa3, wl3, x1, xl1)
        }
      }
#undef SP_LDW
    }
#undef SP_LDX
#undef SP_LDXL
#undef SP_MMA
#undef SP_MMAL
  }

  STAMP(3);
  // ---- epilogue -------------------------------------------------------------------------------------------
  TOUT* __restrict__ yout = reinterpret_cast<TOUT*>(a.y) + (size_t)b * a.YD * a.YH * a.YW * a.CPo;
  const int ox = ox0 + lv;
  int obase[MT];     // element offset of the voxel's channel 0, or -1 when the row/voxel is outside the output
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = wave * MT + m;
    const int rz = r / a.TH, ry = r - rz * a.TH;
    const int oz = oz0 + rz, oy = oy0 + ry;
    const bool valid = oz < a.Do && oy < a.Ho && ox < a.Wo;
    obase[m] = valid ? (((oz * a.osD + a.ooD) * a.YH + (oy * a.osH + a.ooH)) * a.YW + (ox * a.osW + a.ooW)) * a.CPo : -1;
  }
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int c0 = (nt0 + n) * 16 + lg * 4;
    float bj[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) { const float4 bb = *reinterpret_cast<const float4*>(a.bias + c0); bj[0] = bb.x; bj[1] = bb.y; bj[2] = bb.z; bj[3] = bb.w; }
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[n][j] = s2[n][j] = 0.f;
    const bool cok = c0 < a.CPo;
    const bool full = c0 + 4 <= a.Cout;        // partial channel tiles take the masked path
    const bool lin = full && (a.act == SP_ACT_LEAKY || a.act == SP_ACT_NONE);
    const float slope = a.act == SP_ACT_LEAKY ? a.act_param : 1.f;
    const bool want_stats = a.stats != nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v[4];
      if (lin) {
        // leaky(z) = max(z, slope*z) for 0 <= slope <= 1 ; identity is slope = 1
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float z = acc[n][m][j] + bj[j]; v[j] = fmaxf(z, slope * z); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = act_fwd(a.act, a.act_param, acc[n][m][j] + bj[j]);
          v[j] = (c0 + j < a.Cout) ? z : 0.f;
        }
      }
      if (obase[m] >= 0 && cok) {
#ifdef SP_NO_STORE
        if (v[0] == 123456.f)
#endif
        if constexpr (HL) sp_hl_st4(yout + (size_t)obase[m] + c0, a.y_lo_delta, v);
        else Store<TOUT>::st4(yout + (size_t)obase[m] + c0, v);
        if (want_stats) {
          if (sizeof(TOUT) == 2 && !HL) {   // statistics of what is stored (pairs: of the fp32 value)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = bf2f(f2bf(v[j]));
          }
          if (a.stats_mode == 1) {   // BatchNorm-backward sums: (sum g, sum g*x), x read at the same position
            float xv[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!HL) Store<TOUT>::ld4(reinterpret_cast<const TOUT*>(a.aux) + (size_t)b * a.YD * a.YH * a.YW * a.CPo + (size_t)obase[m] + c0, xv);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[n][j] += v[j]; s2[n][j] = fmaf(v[j], xv[j], s2[n][j]); }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[n][j] += v[j]; s2[n][j] = fmaf(v[j], v[j], s2[n][j]); }
          }
        }
      }
    }
  }
  STAMP(4);
  if (a.stats) {
    __syncthreads();
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
    // spread the global accumulators over 8 replicas (by workgroup) to keep same-address atomics apart
    // batched passes (group_batch > 0): the tile's sample decides the BatchNorm group whose rows receive the sums
    double* const g_stats = a.stats + (size_t)(a.group_batch > 0 ? b / a.group_batch : 0) * a.stats_nrep * a.CPo * 2;
    for (int i = tid; i < NT * 16 * 2; i += 256) {
      const int c = nt0 * 16 + (i >> 1);
      if (c < a.CPo) atomicAdd(&g_stats[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (i & 1)], (double)sp_cols_sum(red, NT * 32, 4, i));
    }
  }
  STAMP(5);
}

// ------------------------------------------------------------------------------------------------------------
// Persistent variant for single-group layers with resident weights (the Cout = 16, Cin <= 16 layers that are
// HBM/LDS- rather than MFMA-bound): a workgroup walks tiles blockIdx.x, +gridDim.x, ... with TWO LDS tile buffers.
// Per workgroup ONCE: ktab, weight fragments, per-lane DMA job table.  Per tile: the DMA of tile t+1 is issued
// before the K loop of tile t and lands behind it and the epilogue; BatchNorm statistics stay in registers until
// the single flush at the end (a few hundred atomics per launch instead of ~500 K).
template <int MT, int KS, typename TOUT>
__global__ __launch_bounds__(256) void conv_igemm_persist_kernel(const ConvDmaDev P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_conv_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lv = lane & 15, lg = lane >> 4;

  int* ktab_l = reinterpret_cast<int*>(lds);
  const int ktab_bytes = (KS * 16 + 15) & ~15;
  unsigned char* tile0 = lds + ktab_bytes;
  for (int i = tid; i < KS * 4; i += 256) ktab_l[i] = a.ktab[i];

  int vbase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = wave * MT + m;
    const int rz = r / a.TH, ry = r - rz * a.TH;
    vbase[m] = ((rz * a.sD * a.ITH + ry * a.sH) * a.ITW + lv * a.sW) * a.vsb;
  }
  const bf16x8* __restrict__ wf_hi = reinterpret_cast<const bf16x8*>(a.wfrag_hi);
  bf16x8 wreg[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) wreg[s] = wf_hi[(size_t)s * a.NTtot * 64 + lane];
  const int opp_mask = a.opp - 1;

  // tile-invariant DMA job table (one (plane,row) job per lane, njobs <= 128, one 64-chunk segment per row)
  int job_goff[2], job_loff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int j = k * 64 + lane;
    j = j < P.njobs ? j : P.njobs - 1;
    const uint32_t pl = fdiv(j, P.d_rows);
    const int row = j - pl * P.nrows;
    const uint32_t vz = fdiv(row, P.d_ith);
    const int vy = row - vz * a.ITH;
    job_goff[k] = (((int)vz * a.Hi + vy) * a.Wi) * a.CPi + (int)pl * a.opp * 8;
    job_loff[k] = (int)pl * a.plane_bytes + row * P.row_chunks * 16;
  }
  const int lane_goff = (lane >> P.log2_opp) * a.CPi + (lane & opp_mask) * 8;     // x / octet part of a row chunk
  const int c0 = lg * 4;                                                             // NT == 1
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) { const float4 bb = *reinterpret_cast<const float4*>(a.bias + c0); bj[0] = bb.x; bj[1] = bb.y; bj[2] = bb.z; bj[3] = bb.w; }
  const bool cok = c0 < a.CPo;
  const bool lin = (c0 + 4 <= a.Cout) && (a.act == SP_ACT_LEAKY || a.act == SP_ACT_NONE);
  const float slope = a.act == SP_ACT_LEAKY ? a.act_param : 1.f;
  const bool want_stats = a.stats != nullptr;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};

  struct TileCo { int b, oz0, oy0, ox0, iz0, iy0, ix0; bool interior, border; };
  auto decode = [&](uint32_t tile) {
    TileCo c;
    uint32_t t = tile;
    uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
    q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; t = q;
    q = fdiv(t, P.d_tz); const int tz = t - q * P.ntz; c.b = q;
    c.oz0 = tz * a.TD; c.oy0 = ty * a.TH; c.ox0 = tx * 16;
    c.iz0 = c.oz0 * a.sD + a.o0D; c.iy0 = c.oy0 * a.sH + a.o0H; c.ix0 = c.ox0 * a.sW + a.o0W;
    c.interior = c.iz0 >= 0 && c.iy0 >= 0 && c.ix0 >= 0 && c.iz0 + a.ITD <= a.Di && c.iy0 + a.ITH <= a.Hi && c.ix0 + a.ITW <= a.Wi;
    c.border = a.zfill && !c.interior;
    return c;
  };
  auto issue = [&](const TileCo& c, unsigned char* tile) {
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(a.x) + (size_t)c.b * a.Di * a.Hi * a.Wi * a.CPi;
    if (c.interior) {
      const bf16_t* org = xin + (((size_t)c.iz0 * a.Hi + c.iy0) * a.Wi + c.ix0) * a.CPi;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int nj = min(64, P.njobs - k * 64);
        for (int j = wave; j < nj; j += 4) {
          const int goff = __builtin_amdgcn_readlane(job_goff[k], j);
          const int lo = __builtin_amdgcn_readlane(job_loff[k], j);
          if (lane < P.row_chunks)
            sp_dma16(org + goff + lane_goff, tile + lo);
        }
      }
    } else {
      // tiles touching the volume border: clamp every coordinate (the zero-fill pass fixes padded voxels)
      for (int j = wave; j < P.njobs; j += 4) {
        const int sj = __builtin_amdgcn_readfirstlane(j);
        const uint32_t pl = fdiv(sj, P.d_rows);
        const int row = sj - pl * P.nrows;
        const uint32_t vz = fdiv(row, P.d_ith);
        const int vy = row - vz * a.ITH;
        const int cz = min(max(c.iz0 + (int)vz, 0), a.Di - 1), cy = min(max(c.iy0 + vy, 0), a.Hi - 1);
        if (lane < P.row_chunks) {
          const int cx = min(max(c.ix0 + (lane >> P.log2_opp), 0), a.Wi - 1);
          const bf16_t* src = xin + (((size_t)cz * a.Hi + cy) * a.Wi + cx) * a.CPi + ((int)pl * a.opp + (lane & opp_mask)) * 8;
          sp_dma16(src, tile + (int)pl * a.plane_bytes + row * P.row_chunks * 16);
        }
      }
    }
  };

  // consecutive tile ids on one XCD (halo re-reads hit that XCD's L2): blocks b, b+8, ... share an XCD
  const uint32_t vb = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t tile = vb;
  int cur = 0;
  __syncthreads();
  TileCo tc = decode(tile < P.nblk ? tile : 0);
  if (tile < P.nblk) issue(tc, tile0);
  for (; tile < P.nblk; tile += gridDim.x) {
    unsigned char* tl = tile0 + cur * P.buf_stride;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                     // tile `tile` has landed for everyone; buffer cur^1 is free again
    const uint32_t nxt = tile + gridDim.x;
    TileCo tn = tc;
    if (nxt < P.nblk) { tn = decode(nxt); issue(tn, tile0 + (cur ^ 1) * P.buf_stride); }
    if (tc.border) {
      for (int r = wave; r < P.nruns; r += 4) {
        const int i = r * 64 + lane;
        if (i < P.group_chunks) {
          const uint32_t pl = fdiv(i, P.d_plane);
          const int ii = i - pl * P.plane_chunks;
          const uint32_t vox = (uint32_t)ii >> P.log2_opp;
          const uint32_t row = fdiv(vox, P.d_itw);
          const int vx = vox - row * a.ITW;
          const uint32_t vz = fdiv(row, P.d_ith);
          const int vy = row - vz * a.ITH;
          const int gz = tc.iz0 + (int)vz, gy = tc.iy0 + vy, gx = tc.ix0 + vx;
          if (!((unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi))
            *reinterpret_cast<uint4*>(tl + (size_t)i * 16) = make_uint4(0, 0, 0, 0);
        }
      }
      __syncthreads();
    }
    // ---- K loop (static ping-pong of the activation fragments) -------------------------------------------------
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 x0[MT], x1[MT];
    {
      const int k0 = ktab_l[lg];
#pragma unroll
      for (int m = 0; m < MT; ++m) x0[m] = *reinterpret_cast<const bf16x8*>(tl + vbase[m] + k0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) {
        const int kn = ktab_l[(s + 1) * 4 + lg];
        if ((s & 1) == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m) x1[m] = *reinterpret_cast<const bf16x8*>(tl + vbase[m] + kn);
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) x0[m] = *reinterpret_cast<const bf16x8*>(tl + vbase[m] + kn);
        }
      }
      if ((s & 1) == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = SP_MFMA16(wreg[s], x0[m], acc[m], 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = SP_MFMA16(wreg[s], x1[m], acc[m], 0, 0, 0);
      }
    }
    // ---- epilogue ------------------------------------------------------------------------------------------------
    TOUT* __restrict__ yout = reinterpret_cast<TOUT*>(a.y) + (size_t)tc.b * a.YD * a.YH * a.YW * a.CPo;
    const int ox = tc.ox0 + lv;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int r = wave * MT + m;
      const int rz = r / a.TH, ry = r - rz * a.TH;
      const int oz = tc.oz0 + rz, oy = tc.oy0 + ry;
      const bool valid = oz < a.Do && oy < a.Ho && ox < a.Wo && cok;
      float v[4];
      if (lin) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float z = acc[m][j] + bj[j]; v[j] = fmaxf(z, slope * z); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = act_fwd(a.act, a.act_param, acc[m][j] + bj[j]);
          v[j] = (c0 + j < a.Cout) ? z : 0.f;
        }
      }
      if (valid) {
        const size_t off = (size_t)((((oz * a.osD + a.ooD) * a.YH + (oy * a.osH + a.ooH)) * a.YW + (ox * a.osW + a.ooW)) * a.CPo) + c0;
        Store<TOUT>::st4(yout + off, v);
        if (want_stats) {
          if (sizeof(TOUT) == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = bf2f(f2bf(v[j]));
          }
          if (a.stats_mode == 1) {
            float xv[4];
            Store<TOUT>::ld4(reinterpret_cast<const TOUT*>(a.aux) + (size_t)tc.b * a.YD * a.YH * a.YW * a.CPo + off, xv);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], xv[j], s2[j]); }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], v[j], s2[j]); }
          }
        }
      }
    }
    tc = tn;
    cur ^= 1;
  }
  if (want_stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);      // [4 waves][32] (ordered sum: sp_cols_sum)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x1s = row16_sum(s1[j]), x2s = row16_sum(s2[j]);
      if (lv == 0) { red[wave * 32 + (lg * 4 + j) * 2] = x1s; red[wave * 32 + (lg * 4 + j) * 2 + 1] = x2s; }
    }
    __syncthreads();
    for (int i = tid; i < 32; i += 256) {
      const int c = i >> 1;
      if (c < a.CPo) atomicAdd(&a.stats[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (i & 1)], (double)sp_cols_sum(red, 32, 4, i));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// z-marching variant for the 16-channel-in, 16-channel-out stride-1 layers (the largest volumes of the U-Net):
// a workgroup owns a column of 32 output rows x 16 voxels and walks it in z.  The input lives in a RING of four
// z-plane slots ((32 + kH - 1) x (15 + kW) voxels x 32 B each): per output plane exactly ONE new input plane is
// fetched (halo 1.2x instead of 2.1x per tile of the kernel above), its DMA is issued a whole step ahead, the
// weight fragments and the per-lane DMA plan are set up once per workgroup, the BatchNorm statistics are flushed
// once.  ktab entries of this variant: in-plane byte offset | dz (low two bits).
struct ConvZsDev {
  sp_conv_args a;
  int32_t ITH, ITW, S, nchunks, ZC, nzc, nty, ntx;
  uint32_t ncols;
  FastDiv d_itw, d_tx, d_ty, d_zc;
};

template <int NT, int MT, int KS, typename TOUT>
__global__ __launch_bounds__(256, (NT == 1 ? 2 : 1)) void conv_igemm_zs_kernel(const ConvZsDev P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int NJ = 5;                                   // 16-byte chunks of one plane per lane (host-checked)
  const sp_conv_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lv = lane & 15, lg = lane >> 4;
  unsigned char* ring = lds;

  int kv[KS];                                             // this lane group's (in-plane offset | dz) per K step
#pragma unroll
  for (int s = 0; s < KS; ++s) kv[s] = a.ktab[s * 4 + lg];
  int vbase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) vbase[m] = ((wave * MT + m) * P.ITW + lv) * 32;
  const bf16x8* __restrict__ wf_hi = reinterpret_cast<const bf16x8*>(a.wfrag_hi);
  bf16x8 wreg[KS][NT];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n) wreg[s][n] = wf_hi[((size_t)s * a.NTtot + n) * 64 + lane];

  // per-lane DMA plan of one plane: chunk c = (wave + 4j)*64 + lane -> (row vy, voxel vx, half)
  uint32_t rel[NJ];
  int crd[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;
    const int cc = c < P.nchunks ? c : P.nchunks - 1;
    const int half = cc & 1, vox = cc >> 1;
    const int vy = fdiv(vox, P.d_itw), vx = vox - vy * P.ITW;
    rel[j] = (uint32_t)(((vy * a.Wi + vx) * a.CPi + half * 8) * 2);
    crd[j] = vy | (vx << 8) | (c < P.nchunks ? 0 : (1 << 30));       // bit 30: no such chunk (tail of the last round)
  }
  float bj[NT][4], s1[NT][4], s2[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bj[n][j] = a.bias ? a.bias[n * 16 + lg * 4 + j] : 0.f; s1[n][j] = s2[n][j] = 0.f; }
  const bool linact = a.act == SP_ACT_LEAKY || a.act == SP_ACT_NONE;
  const float slope = a.act == SP_ACT_LEAKY ? a.act_param : 1.f;
  const bool want_stats = a.stats != nullptr;

  // Work = (column, output plane) pairs cut into gridDim.x equal pieces of the flattened sequence (perfect balance for
  // any volume; a piece that crosses a column boundary pays one more three-plane prologue).  XCD-aware piece id.
  const uint32_t vb = xcd_remap(blockIdx.x, gridDim.x);
  const uint64_t T = (uint64_t)P.ncols * a.Do;
  uint64_t pos = T * vb / gridDim.x;
  const uint64_t pend = T * (vb + 1) / gridDim.x;
  while (pos < pend) {
    const uint32_t col = (uint32_t)(pos / (uint32_t)a.Do);
    const int z0 = (int)(pos - (uint64_t)col * a.Do);
    const int z1 = (int)min((uint64_t)a.Do, (uint64_t)z0 + (pend - pos));
    pos += (uint64_t)(z1 - z0);
    uint32_t t = col;
    uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
    q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; const int b = q;
    const int oy0 = ty * (4 * MT), ox0 = tx * 16;
    const int iy0 = oy0 + a.o0H, ix0 = ox0 + a.o0W;
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * a.CPi;
    // in-plane validity of this lane's chunks (column-invariant): bit j set = inside the volume
    int vmask = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int vy = crd[j] & 0xff, vx = (crd[j] >> 8) & 0xff;
      if (!(crd[j] >> 30) && (unsigned)(iy0 + vy) < (unsigned)a.Hi && (unsigned)(ix0 + vx) < (unsigned)a.Wi) vmask |= 1 << j;
    }
    auto load_plane = [&](int iz, int slot) {
      unsigned char* dst0 = ring + slot * P.S;
      const bool zin = (unsigned)iz < (unsigned)a.Di;
      const unsigned char* src0 = reinterpret_cast<const unsigned char*>(xin) + (((int64_t)iz * a.Hi + iy0) * a.Wi + ix0) * a.CPi * 2;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        unsigned char* dst = dst0 + (wave + 4 * j) * 1024;
        if (!(crd[j] >> 30)) {
          if (zin && ((vmask >> j) & 1)) sp_dma16(src0 + rel[j], dst);
          else *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(0, 0, 0, 0);      // padding
        }
      }
    };
    __syncthreads();                                      // the previous column has been consumed
    load_plane(z0 + a.o0D, 0);
    load_plane(z0 + a.o0D + 1, 1);
    load_plane(z0 + a.o0D + 2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TOUT* __restrict__ yout = reinterpret_cast<TOUT*>(a.y) + (size_t)b * a.YD * a.YH * a.YW * a.CPo;
    const int ox = ox0 + lv;
    for (int z = z0; z < z1; ++z) {
      const int sl = (z - z0) & 3;
      if (z + 1 < z1) load_plane(z + a.o0D + 3, (sl + 3) & 3);         // lands behind this step's MFMAs
      // ---- K loop over the three resident planes
      f32x4 acc[NT][MT];
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 x0[MT], x1[MT];
#define ZS_OFF(s_) ((((sl + (kv[s_] & 3)) & 3) * P.S) + (kv[s_] & ~15))
      {
        const int k0 = ZS_OFF(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) x0[m] = *reinterpret_cast<const bf16x8*>(ring + vbase[m] + k0);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
          const int kn = ZS_OFF(s + 1 < KS ? s + 1 : s);
          if ((s & 1) == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) x1[m] = *reinterpret_cast<const bf16x8*>(ring + vbase[m] + kn);
          } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) x0[m] = *reinterpret_cast<const bf16x8*>(ring + vbase[m] + kn);
          }
        }
        if ((s & 1) == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n][m] = SP_MFMA16(wreg[s][n], x0[m], acc[n][m], 0, 0, 0);
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n][m] = SP_MFMA16(wreg[s][n], x1[m], acc[n][m], 0, 0, 0);
        }
      }
#undef ZS_OFF
      // ---- epilogue of output plane z
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + wave * MT + m;
        const bool inside = oy < a.Ho && ox < a.Wo;
        const size_t vo = (size_t)((((z * a.osD + a.ooD) * a.YH + (oy * a.osH + a.ooH)) * a.YW + (ox * a.osW + a.ooW)) * a.CPo);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int c0 = n * 16 + lg * 4;
          const bool lin = linact && (c0 + 4 <= a.Cout);
          float v[4];
          if (lin) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float zz = acc[n][m][j] + bj[n][j]; v[j] = fmaxf(zz, slope * zz); }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float zz = act_fwd(a.act, a.act_param, acc[n][m][j] + bj[n][j]);
              v[j] = (c0 + j < a.Cout) ? zz : 0.f;
            }
          }
          if (inside && c0 < a.CPo) {
            Store<TOUT>::st4(yout + vo + c0, v);
            if (want_stats) {
              if (sizeof(TOUT) == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = bf2f(f2bf(v[j]));
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) { s1[n][j] += v[j]; s2[n][j] = fmaf(v[j], v[j], s2[n][j]); }
            }
          }
        }
      }
#ifndef SP_ZS_NOWAIT       // diagnostic variant (tools/build_variant.sh): wrong results, shows what the DMA wait costs
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the plane issued at the top of this step has landed
#endif
      __syncthreads();                                    // and every wave is done with the oldest slot
    }
  }
  if (want_stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);      // [4 waves][NT * 32] (ordered sum: sp_cols_sum)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x1s = row16_sum(s1[n][j]), x2s = row16_sum(s2[n][j]);
        if (lv == 0) { red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2] = x1s; red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2 + 1] = x2s; }
      }
    __syncthreads();
    for (int i = tid; i < NT * 32; i += 256) {
      const int c = i >> 1;
      if (c < a.CPo) atomicAdd(&a.stats[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (i & 1)], (double)sp_cols_sum(red, NT * 32, 4, i));
    }
  }
}

// ROW-REUSE variant of the z-marching kernel for one output tile (Cout <= 16), 3x3x3, one 16-channel input plane.
// With 16 output channels every activation fragment read from LDS (1 KB) feeds ONE MFMA in the kernels above, which
// makes them LDS-bandwidth bound at <= 50 % of the matrix pipe.  Here the K steps are ordered so that both taps of a
// step share dy (5 steps per dy, 15 in all, one half-step of zero weights): the fragment of input row r and step type t
// then serves the output rows r, r-1, r-2 (dy = 0, 1, 2) -- a wave owns 8 consecutive output rows, loads the 10 x 5
// fragments of its input rows once and issues 120 MFMAs from them: 2.4 MFMAs per LDS read instead of 1.
// ktab = 5 x 4 entries (in-plane offset of the step type at dy = 0 | dz); weights: 15 fragments in (dy, type) order.
template <typename TOUT>
__global__ __launch_bounds__(256, 2) void conv_igemm_zr_kernel(const ConvZsDev P) {
  constexpr int NT = 1, MT = 8, KS = 15, NTY = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int NJ = 5;                                   // 16-byte chunks of one plane per lane (host-checked)
  const sp_conv_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lv = lane & 15, lg = lane >> 4;
  unsigned char* ring = lds;

  int kv[NTY];                                            // this lane group's (in-plane offset at dy = 0 | dz) per step type
#pragma unroll
  for (int t = 0; t < NTY; ++t) kv[t] = a.ktab[t * 4 + lg];
  int vbase[MT + 2];                                      // the wave's ten input rows
#pragma unroll
  for (int r = 0; r < MT + 2; ++r) vbase[r] = ((wave * MT + r) * P.ITW + lv) * 32;
  const bf16x8* __restrict__ wf_hi = reinterpret_cast<const bf16x8*>(a.wfrag_hi);
  bf16x8 wreg[KS][NT];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n) wreg[s][n] = wf_hi[((size_t)s * a.NTtot + n) * 64 + lane];

  // per-lane DMA plan of one plane: chunk c = (wave + 4j)*64 + lane -> (row vy, voxel vx, half)
  uint32_t rel[NJ];
  int crd[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;
    const int cc = c < P.nchunks ? c : P.nchunks - 1;
    const int half = cc & 1, vox = cc >> 1;
    const int vy = fdiv(vox, P.d_itw), vx = vox - vy * P.ITW;
    rel[j] = (uint32_t)(((vy * a.Wi + vx) * a.CPi + half * 8) * 2);
    crd[j] = vy | (vx << 8) | (c < P.nchunks ? 0 : (1 << 30));       // bit 30: no such chunk (tail of the last round)
  }
  float bj[NT][4], s1[NT][4], s2[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bj[n][j] = a.bias ? a.bias[n * 16 + lg * 4 + j] : 0.f; s1[n][j] = s2[n][j] = 0.f; }
  const bool linact = a.act == SP_ACT_LEAKY || a.act == SP_ACT_NONE;
  const float slope = a.act == SP_ACT_LEAKY ? a.act_param : 1.f;
  const bool want_stats = a.stats != nullptr;

  // Work = (column, output plane) pairs cut into gridDim.x equal pieces of the flattened sequence (perfect balance for
  // any volume; a piece that crosses a column boundary pays one more three-plane prologue).  XCD-aware piece id.
  const uint32_t vb = xcd_remap(blockIdx.x, gridDim.x);
  const uint64_t T = (uint64_t)P.ncols * a.Do;
  uint64_t pos = T * vb / gridDim.x;
  const uint64_t pend = T * (vb + 1) / gridDim.x;
  while (pos < pend) {
    const uint32_t col = (uint32_t)(pos / (uint32_t)a.Do);
    const int z0 = (int)(pos - (uint64_t)col * a.Do);
    const int z1 = (int)min((uint64_t)a.Do, (uint64_t)z0 + (pend - pos));
    pos += (uint64_t)(z1 - z0);
    uint32_t t = col;
    uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
    q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; const int b = q;
    const int oy0 = ty * (4 * MT), ox0 = tx * 16;
    const int iy0 = oy0 + a.o0H, ix0 = ox0 + a.o0W;
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * a.CPi;
    // in-plane validity of this lane's chunks (column-invariant): bit j set = inside the volume
    int vmask = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int vy = crd[j] & 0xff, vx = (crd[j] >> 8) & 0xff;
      if (!(crd[j] >> 30) && (unsigned)(iy0 + vy) < (unsigned)a.Hi && (unsigned)(ix0 + vx) < (unsigned)a.Wi) vmask |= 1 << j;
    }
    auto load_plane = [&](int iz, int slot) {
      unsigned char* dst0 = ring + slot * P.S;
      const bool zin = (unsigned)iz < (unsigned)a.Di;
      const unsigned char* src0 = reinterpret_cast<const unsigned char*>(xin) + (((int64_t)iz * a.Hi + iy0) * a.Wi + ix0) * a.CPi * 2;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        unsigned char* dst = dst0 + (wave + 4 * j) * 1024;
        if (!(crd[j] >> 30)) {
          if (zin && ((vmask >> j) & 1)) sp_dma16(src0 + rel[j], dst);
          else *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(0, 0, 0, 0);      // padding
        }
      }
    };
    __syncthreads();                                      // the previous column has been consumed
    load_plane(z0 + a.o0D, 0);
    load_plane(z0 + a.o0D + 1, 1);
    load_plane(z0 + a.o0D + 2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TOUT* __restrict__ yout = reinterpret_cast<TOUT*>(a.y) + (size_t)b * a.YD * a.YH * a.YW * a.CPo;
    const int ox = ox0 + lv;
    // The stores of plane z are issued at the top of iteration z + 1 (after the next plane's DMA): the s_waitcnt vmcnt(0)
    // that closes an iteration also waits for outstanding STORES on gfx9, and issued right before it their whole latency
    // was exposed once per plane; one K loop later they have long been acknowledged.
    float pend[MT][4];
    auto flush = [&](int zp) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + wave * MT + m;
        const int c0 = lg * 4;
        if (oy < a.Ho && ox < a.Wo && c0 < a.CPo) {
          const size_t vo = (size_t)((((zp * a.osD + a.ooD) * a.YH + (oy * a.osH + a.ooH)) * a.YW + (ox * a.osW + a.ooW)) * a.CPo);
          Store<TOUT>::st4(yout + vo + c0, pend[m]);
        }
      }
    };
    for (int z = z0; z < z1; ++z) {
      const int sl = (z - z0) & 3;
      if (z + 1 < z1) load_plane(z + a.o0D + 3, (sl + 3) & 3);         // lands behind this step's MFMAs
      if (z > z0) flush(z - 1);
      // ---- K loop over the three resident planes
      f32x4 acc[NT][MT];
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 xr[2][NTY];                                    // the five fragments of an input row, double-buffered by row
#define ZS_OFF(t_) ((((sl + (kv[t_] & 3)) & 3) * P.S) + (kv[t_] & ~15))
#pragma unroll
      for (int t = 0; t < NTY; ++t) xr[0][t] = *reinterpret_cast<const bf16x8*>(ring + vbase[0] + ZS_OFF(t));
#pragma unroll
      for (int r = 0; r < MT + 2; ++r) {
        if (r + 1 < MT + 2) {                                 // next row's fragments: 15 MFMAs (240 cycles) cover their latency
#pragma unroll
          for (int t = 0; t < NTY; ++t) xr[(r + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(ring + vbase[r + 1] + ZS_OFF(t));
        }
#pragma unroll
        for (int t = 0; t < NTY; ++t) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int m = r - dy;                             // output row served through tap row dy
            if (m >= 0 && m < MT) acc[0][m] = SP_MFMA16(wreg[dy * NTY + t][0], xr[r & 1][t], acc[0][m], 0, 0, 0);
          }
        }
      }
#undef ZS_OFF
      // ---- epilogue of output plane z
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + wave * MT + m;
        const bool inside = oy < a.Ho && ox < a.Wo;
        const size_t vo = (size_t)((((z * a.osD + a.ooD) * a.YH + (oy * a.osH + a.ooH)) * a.YW + (ox * a.osW + a.ooW)) * a.CPo);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int c0 = n * 16 + lg * 4;
          const bool lin = linact && (c0 + 4 <= a.Cout);
          float v[4];
          if (lin) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float zz = acc[n][m][j] + bj[n][j]; v[j] = fmaxf(zz, slope * zz); }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float zz = act_fwd(a.act, a.act_param, acc[n][m][j] + bj[n][j]);
              v[j] = (c0 + j < a.Cout) ? zz : 0.f;
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) pend[m][j] = v[j];
          if (inside && c0 < a.CPo) {
            if (want_stats) {
              if (sizeof(TOUT) == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = bf2f(f2bf(v[j]));
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) { s1[n][j] += v[j]; s2[n][j] = fmaf(v[j], v[j], s2[n][j]); }
            }
          }
        }
      }
#ifndef SP_ZS_NOWAIT       // diagnostic variant (tools/build_variant.sh): wrong results, shows what the DMA wait costs
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the plane issued at the top of this step has landed
#endif
      __syncthreads();                                    // and every wave is done with the oldest slot
    }
    if (z1 > z0) flush(z1 - 1);
  }
  if (want_stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);      // [4 waves][NT * 32] (ordered sum: sp_cols_sum)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x1s = row16_sum(s1[n][j]), x2s = row16_sum(s2[n][j]);
        if (lv == 0) { red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2] = x1s; red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2 + 1] = x2s; }
      }
    __syncthreads();
    for (int i = tid; i < NT * 32; i += 256) {
      const int c = i >> 1;
      if (c < a.CPo) atomicAdd(&a.stats[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (i & 1)], (double)sp_cols_sum(red, NT * 32, 4, i));
    }
  }
}

static int launch_zs(const sp_conv_args* a, hipStream_t st) {
  SP_CHECK_ARG(a->NT == a->NTtot && a->NT >= 1 && a->NT <= 3 && a->ngroups == 1 && a->opp == 2 && a->vsb == 32 && a->octs_per_group == 2,
               "sp_conv3d_igemm(zs): one 16-channel plane in, up to three 16-channel tiles out");
  SP_CHECK_ARG(a->sD == 1 && a->sH == 1 && a->sW == 1 && a->stats_mode == 0 && a->in_scale == nullptr && a->dtype_in == SP_BF16,
               "sp_conv3d_igemm(zs): stride 1, plain statistics, bf16 input without affine on load");
  const bool zr = a->persist == 4;                     // row-reuse variant: 15 weight fragments in (dy, type) order, 5 x 4 table
  SP_CHECK_ARG(zr ? (a->NT == 1 && a->ITH_zs == 34) : (a->steps_per_group == 14 || (a->steps_per_group == 7 && a->NT == 1)),
               "sp_conv3d_igemm(zs): 14 (or 7) resident K steps; row-reuse: one output tile, 3x3x3");
  ConvZsDev P;
  P.a = *a;
  const int MT = a->NT == 3 ? 4 : 8;                   // rows per wave: three resident weight tiles leave room for 4
  P.ITH = 4 * MT + (a->ITH_zs - 32); P.ITW = a->ITW;   // ITH_zs = 32 + kernel extent - 1 (planner)
  SP_CHECK_ARG(a->ITH_zs >= 32 && a->ITH_zs <= 40 && P.ITW >= 16 && P.ITW <= 24, "sp_conv3d_igemm(zs): plane extent");
  P.S = P.ITH * P.ITW * 32;
  P.nchunks = P.ITH * P.ITW * 2;
  SP_CHECK_ARG(P.nchunks <= 5 * 256 && 4 * P.S <= 160 * 1024, "sp_conv3d_igemm(zs): plane does not fit the per-lane plan");
  P.ntx = (a->Wo + 15) / 16; P.nty = (a->Ho + 4 * MT - 1) / (4 * MT);
  const int cols_xy = a->B * P.nty * P.ntx;
  static const int zs_mult_ = getenv("SP_ZS_SLOTS") ? atoi(getenv("SP_ZS_SLOTS")) : 1;
  const int slots = (a->NT == 1 ? 512 : 256) * zs_mult_;            // resident workgroups (2 / 1 per CU)
  P.nzc = 1; P.ZC = a->Do;
  P.ncols = (uint32_t)cols_xy;
  P.d_itw = make_fastdiv(P.ITW); P.d_tx = make_fastdiv(P.ntx); P.d_ty = make_fastdiv(P.nty); P.d_zc = make_fastdiv(P.nzc);
  const int lds_bytes = 4 * P.S;
  const uint64_t planes = (uint64_t)cols_xy * a->Do;
  const unsigned grid = planes / 4 < (uint64_t)slots ? (unsigned)(planes / 4 > 0 ? planes / 4 : 1) : (unsigned)slots;   // >= 4 planes per piece
#define SP_ZS(N_, M_, K_, T_)                                                                                        \
  {                                                                                                                  \
    auto kern = conv_igemm_zs_kernel<N_, M_, K_, T_>;                                                                \
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv");                                                                       \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, st, P);                                               \
  }
#define SP_ZS_T(N_, M_, K_) { if (a->dtype_out == SP_F32) SP_ZS(N_, M_, K_, float) else SP_ZS(N_, M_, K_, bf16_t) }
  if (zr) {
    if (a->dtype_out == SP_F32) {
      auto kern = conv_igemm_zr_kernel<float>;
      SP_ENSURE_LDS(kern, lds_bytes, "sp_conv");
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, st, P);
    } else {
      auto kern = conv_igemm_zr_kernel<bf16_t>;
      SP_ENSURE_LDS(kern, lds_bytes, "sp_conv");
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, st, P);
    }
  } else if (a->NT == 1 && a->steps_per_group == 14) SP_ZS_T(1, 8, 14)
  else if (a->NT == 1) SP_ZS_T(1, 8, 7)
  else if (a->NT == 2) SP_ZS_T(2, 8, 14)
  else SP_ZS_T(3, 4, 14)
#undef SP_ZS_T
#undef SP_ZS
  SP_CHECK_LAUNCH("sp_conv3d_igemm(zs)");
  return SP_OK;
}

template <int MT, int KS>
static int launch_persist(ConvDmaDev& P, hipStream_t st) {
  const int buf = (((P.nruns * 1024) + 1023) / 1024) * 1024;
  P.buf_stride = buf;
  const int lds_bytes = ((KS * 16 + 15) & ~15) + 2 * buf;
  const unsigned grid = P.nblk < 512u ? P.nblk : 512u;
#define SP_PL(T_)                                                                                                  \
  {                                                                                                                \
    auto kern = conv_igemm_persist_kernel<MT, KS, T_>;                                                             \
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv");                                                          \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, st, P);                                             \
  }
  if (P.a.dtype_out == SP_F32) SP_PL(float) else SP_PL(bf16_t)
#undef SP_PL
  SP_CHECK_LAUNCH("sp_conv3d_igemm(persist)");
  return SP_OK;
}

template <int NT, int MT, int KS, typename TOUT>
static int launch_dma(const ConvDmaDev& P, dim3 grid, hipStream_t st) {
  auto kern = conv_igemm_dma_kernel<NT, MT, KS, TOUT>;
  SP_ENSURE_LDS(kern, P.a.lds_bytes, "sp_conv3d_igemm");
  hipLaunchKernelGGL(kern, grid, dim3(256), P.a.lds_bytes, st, P);
  SP_CHECK_LAUNCH("sp_conv3d_igemm(dma)");
  return SP_OK;
}

template <int NT, int MT, int KS>
static int launch_out(const ConvDmaDev& P, dim3 grid, hipStream_t st) {
  if (P.a.dtype_out == SP_F32) return launch_dma<NT, MT, KS, float>(P, grid, st);
  return launch_dma<NT, MT, KS, bf16_t>(P, grid, st);
}
// bf16 pairs in and out: the run-time K loop with hi / lo tiles and fragments
template <int NT, int MT>
static int launch_hl(const ConvDmaDev& P, dim3 grid, hipStream_t st) {
  auto kern = conv_igemm_dma_kernel<NT, MT, 0, sp_hl_t, true>;
  SP_ENSURE_LDS(kern, P.a.lds_bytes, "sp_conv3d_igemm");
  hipLaunchKernelGGL(kern, grid, dim3(256), P.a.lds_bytes, st, P);
  SP_CHECK_LAUNCH("sp_conv3d_igemm(dma, pairs)");
  return SP_OK;
}

// resident-weight variants exist for KS*NT <= 16 and KS in {1, 2, 4, 7, 14}
template <int NT, int MT>
static int dispatch_dma(const ConvDmaDev& P, dim3 grid, hipStream_t st) {
  const int ks = P.a.steps_per_group;
#define SP_KS(K_) if (ks == K_ && K_ * NT <= 16) return launch_out<NT, MT, (K_ * NT <= 16 ? K_ : 0)>(P, grid, st)
  SP_KS(1); SP_KS(2); SP_KS(4); SP_KS(7); SP_KS(14);
#undef SP_KS
  if (ks % 2 != 0) { sp_set_error("sp_conv3d_igemm(dma): run-time K loop needs an even step count (got %d)", ks); return SP_EINVAL; }
  return launch_out<NT, MT, 0>(P, grid, st);
}

int sp_conv3d_igemm_dma(const sp_conv_args* a, sp_stream_t stream) {
  const bool hl = a->dtype_in == SP_HL;
  SP_CHECK_ARG((a->dtype_in == SP_BF16 || hl) && a->in_scale == nullptr, "sp_conv3d_igemm(dma): needs bf16 (or bf16-pair) input and no affine on load");
  SP_CHECK_ARG(!hl || (a->dtype_out == SP_HL && a->wfrag_lo && a->lo_offset > 0 && a->x_lo_delta != 0 && a->y_lo_delta != 0 && a->persist == 0 &&
                       a->steps_per_group % 2 == 0 && a->stats_mode == 0 && a->group_batch == 0),
               "sp_conv3d_igemm(dma): bf16 pairs need lo fragments, lo tile offset, lo deltas of x and y, an even step count, the tiled kernel");
  SP_CHECK_ARG(a->group_batch == 0 || (a->persist == 0 && a->group_batch > 0 && a->B % a->group_batch == 0),
               "sp_conv3d_igemm(dma): BatchNorm groups (group_batch %d, B %d) need the tiled kernel (persist 0) and whole groups", a->group_batch, a->B);
  if (a->persist == 3 || a->persist == 4) return launch_zs(a, reinterpret_cast<hipStream_t>(stream));      // z-marching plans (ktab in their format)
  SP_CHECK_ARG(a->x_plane == 0 || (a->opp == 2 && !a->persist && a->x_plane < (1ll << 31)), "sp_conv3d_igemm(dma): plane-major input needs 16-channel planes");
  SP_CHECK_ARG(a->opp == 1 || a->opp == 2, "sp_conv3d_igemm(dma): octets per plane must be 1 or 2");
  SP_CHECK_ARG(a->vsb == a->opp * 16 && a->plane_bytes == a->ITD * a->ITH * a->ITW * a->vsb, "sp_conv3d_igemm(dma): planes must be lane-linear (no padding)");
  SP_CHECK_ARG(a->octs_per_group % a->opp == 0, "sp_conv3d_igemm(dma): group does not consist of whole planes");
  ConvDmaDev P;
  P.a = *a;
  P.log2_opp = a->opp == 2 ? 1 : 0;
  P.plane_chunks = a->ITD * a->ITH * a->ITW * a->opp;
  P.group_chunks = P.plane_chunks * (a->octs_per_group / a->opp);
  P.nruns = (P.group_chunks + 63) / 64;
  P.row_chunks = a->ITW * a->opp;
  P.segs_per_row = (P.row_chunks + 63) / 64;
  P.nrows = a->ITD * a->ITH;
  P.njobs = (a->octs_per_group / a->opp) * P.nrows * P.segs_per_row;
  SP_CHECK_ARG(P.njobs <= 128, "sp_conv3d_igemm(dma): %d DMA jobs per group (max 128)", P.njobs);
  P.d_segs = make_fastdiv(P.segs_per_row);
  P.d_rows = make_fastdiv(P.nrows);
  // (pairs: every DMA job and the zero fill write whole chunks of their own tile only, so the lo tile may start right behind the hi one)
  const long need = ((a->steps_per_group * 16 + 15) & ~15) + (hl ? a->lo_offset + (long)P.group_chunks * 16 : (long)P.nruns * 1024);
  SP_CHECK_ARG(!hl || a->lo_offset >= (long)P.group_chunks * 16, "sp_conv3d_igemm(dma): lo tile offset %d inside the hi tile (%ld bytes)", a->lo_offset, (long)P.group_chunks * 16);
  SP_CHECK_ARG(need <= a->lds_bytes && a->lds_bytes <= 160 * 1024, "sp_conv3d_igemm(dma): LDS plan too small (need %ld, have %d)", need, a->lds_bytes);
  P.d_itw = make_fastdiv(a->ITW);
  P.d_ith = make_fastdiv(a->ITH);
  P.d_plane = make_fastdiv(P.plane_chunks);
  P.ntx = (a->Wo + 15) / 16;
  P.nty = (a->Ho + a->TH - 1) / a->TH;
  P.ntz = (a->Do + a->TD - 1) / a->TD;
  P.d_tx = make_fastdiv(P.ntx);
  P.d_ty = make_fastdiv(P.nty);
  P.d_tz = make_fastdiv(P.ntz);
  const uint64_t nblk = (uint64_t)P.ntx * P.nty * P.ntz * a->B;
  SP_CHECK_ARG(nblk < (1ull << 31), "sp_conv3d_igemm(dma): grid too large");
  P.nblk = (uint32_t)nblk;
  dim3 grid(P.nblk, a->NTtot / a->NT);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // persistent double-buffered variant: single channel group, one cout tile, resident weights, whole rows per DMA
  if (a->persist && a->NT == 1 && a->NTtot == 1 && a->ngroups == 1 && P.segs_per_row == 1 && a->MT == 8 &&
      (a->steps_per_group == 14 || a->steps_per_group == 7) && P.nblk >= (a->persist >= 2 ? 2u : 1024u) &&
      ((a->steps_per_group * 16 + 15) & ~15) + 2 * ((P.nruns * 1024 + 1023) / 1024 * 1024) <= 76 * 1024) {
    if (a->steps_per_group == 14) return launch_persist<8, 14>(P, st);
    return launch_persist<8, 7>(P, st);
  }
  if (hl) {
#define SP_CASE_HL(NT_, MT_) if (a->NT == NT_ && a->MT == MT_) return launch_hl<NT_, MT_>(P, grid, st)
    SP_CASE_HL(1, 8); SP_CASE_HL(2, 8); SP_CASE_HL(4, 8); SP_CASE_HL(1, 4); SP_CASE_HL(2, 4); SP_CASE_HL(4, 4); SP_CASE_HL(2, 2); SP_CASE_HL(4, 2);
#undef SP_CASE_HL
    sp_set_error("sp_conv3d_igemm(dma): no bf16-pair kernel for NT=%d MT=%d", a->NT, a->MT);
    return SP_EINVAL;
  }
#define SP_CASE(NT_, MT_) if (a->NT == NT_ && a->MT == MT_) return dispatch_dma<NT_, MT_>(P, grid, st)
  SP_CASE(1, 8); SP_CASE(2, 8); SP_CASE(3, 8); SP_CASE(4, 8);
  SP_CASE(1, 4); SP_CASE(2, 4); SP_CASE(3, 4); SP_CASE(4, 4);
  SP_CASE(1, 2); SP_CASE(2, 2); SP_CASE(3, 2); SP_CASE(4, 2);
#undef SP_CASE
  sp_set_error("sp_conv3d_igemm(dma): no kernel for NT=%d MT=%d", a->NT, a->MT);
  return SP_EINVAL;
}
