// Weight gradient of the 3-D convolutions (backward of Unet3D.py:19,22 / Cae3D.py:41-218) on gfx950.
//
//   dw[tap][co][ci] += sum_{b,o} dz[b,o,co] * xin[b, o*s + o0 + tapoff(tap), ci]
//
// GEMM view: D[co 16][ci 16] += A[co][k] * B[k][ci] with k = 32 consecutive output voxels along x.
// Both operands are channels-last in memory (k is the SLOW index), so both are staged into LDS in
// 16-channel planes and read with ds_read_b64_tr_b16 (gfx950 transposing LDS read): one read hands
// each lane 4 consecutive voxels of one channel.  The voxel <-> (lane group, element) map inside a
// K step is the same for A and B and is chosen so that the two 16-lane groups of a 32-lane half hit
// different 128-byte halves of the bank row (conflict free at 32 B per voxel).
// Taps are split over the 4 waves (<= 7 each); a workgroup owns COB x CIB (cout, cin) tiles and walks
// output tiles persistently, keeping all its dw partial sums in registers; one fp32 atomic flush
// per workgroup at the end (a few hundred KB per launch: far below the chip-wide atomic rate).
#include "sp_common.h"

struct WgradDev {
  sp_wgrad_args a;
  FastDiv d_tx, d_ty, d_tz, d_xw, d_xh;
  uint32_t ntx, nty, ntz, ntiles;
  int32_t TZ, TY, XD, XH, XW;     // output tile rows; staged x tile extent
  int32_t xplane, dzplane, lo_off;  // bytes
  int32_t mdz, mdy, mdx;          // min tap offsets (tile origin shift)
};

#define WG_TAPS_PER_WAVE 7
#define WG_VSB 32

__device__ __forceinline__ bf16x8 tr_read2(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p0));
  bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p1));
  return __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// stage nvox voxels x 16 channels (one plane) : global channels-last -> (affine) -> bf16 hi/lo -> LDS
template <int NP, typename T>
__device__ __forceinline__ void put_chunk(unsigned char* dst, int lo_off, const float* v) {
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
    *reinterpret_cast<uint4*>(dst + lo_off) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
  }
}

template <int NP, int COB, int CIB, typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradDev P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int co_t0 = blockIdx.y * COB, ci_t0 = blockIdx.z * CIB;

  unsigned char* xt = lds;                                 // [CIB][XV][32 B]
  unsigned char* dzt = lds + CIB * P.xplane;               // [COB][TV][32 B]

  // taps of this wave
  const int tap0 = wave * WG_TAPS_PER_WAVE;
  int ntw = a.ntap - tap0; ntw = ntw < 0 ? 0 : (ntw > WG_TAPS_PER_WAVE ? WG_TAPS_PER_WAVE : ntw);
  int tapoff[WG_TAPS_PER_WAVE];
#pragma unroll
  for (int t = 0; t < WG_TAPS_PER_WAVE; ++t) {
    int o = 0;
    if (t < ntw) {
      const int* tp = a.taps + (tap0 + t) * 3;
      o = (((tp[0] - P.mdz) * P.XH + (tp[1] - P.mdy)) * P.XW + (tp[2] - P.mdx)) * WG_VSB;
    }
    tapoff[t] = o;
  }
  // per-lane voxel-in-run for the two transposed reads (see header): quad = 2g+rd (g even), 2g+1-rd (g odd)
  const int vq0 = ((lg & 1) ? 2 * lg + 1 : 2 * lg) * 4 + lq;
  const int vq1 = ((lg & 1) ? 2 * lg : 2 * lg + 1) * 4 + lq;
  const int a_off0 = vq0 * WG_VSB + lp * 8, a_off1 = vq1 * WG_VSB + lp * 8;
  const int b_off0 = vq0 * a.sW * WG_VSB + lp * 8, b_off1 = vq1 * a.sW * WG_VSB + lp * 8;

  f32x4 acc[WG_TAPS_PER_WAVE][COB][CIB];
#pragma unroll
  for (int t = 0; t < WG_TAPS_PER_WAVE; ++t)
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
      for (int i = 0; i < CIB; ++i) acc[t][c][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int TV = P.TZ * P.TY * 32;
  const int XV = P.XD * P.XH * P.XW;
  const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ dzg = reinterpret_cast<const T*>(a.dz);

  for (uint32_t tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    uint32_t t = tile;
    uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
    q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; t = q;
    q = fdiv(t, P.d_tz); const int tz = t - q * P.ntz; const int b = q;
    const int oz0 = tz * P.TZ, oy0 = ty * P.TY, ox0 = tx * 32;
    const int iz0 = oz0 * a.sD + a.o0D + P.mdz, iy0 = oy0 * a.sH + a.o0H + P.mdy, ix0 = ox0 * a.sW + a.o0W + P.mdx;

    __syncthreads();   // previous tile fully consumed
    // ---- dz tile: TZ*TY rows of 32 voxels, COB planes ------------------------------------------
    for (int i = tid; i < TV * COB * 2; i += 256) {
      const int half = i & 1, rest = i >> 1;
      const int pl = rest / TV, vox = rest - pl * TV;
      const int rx = vox & 31, row = vox >> 5;
      const int rz = row / P.TY, ry = row - rz * P.TY;
      const int oz = oz0 + rz, oy = oy0 + ry, ox = ox0 + rx;
      const int c = (co_t0 + pl) * 16 + half * 8;
      float v[8];
      if (oz < a.Do && oy < a.Ho && ox < a.Wo && c < a.CPo) {
        Store<T>::ld8(dzg + ((((size_t)b * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.CPo + c, v);
        if (a.dz_scale) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a.dz_scale[c + j], a.dz_shift[c + j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      put_chunk<NP, T>(dzt + pl * P.dzplane + vox * WG_VSB + half * 16, P.lo_off, v);
    }
    // ---- x halo tile with BatchNorm-on-load, zero outside the volume ------------------------------
    for (int i = tid; i < XV * CIB * 2; i += 256) {
      const int half = i & 1, rest = i >> 1;
      const int pl = rest / XV;
      const uint32_t vox = rest - pl * XV;
      const uint32_t row = fdiv(vox, P.d_xw);
      const int vx = vox - row * P.XW;
      const uint32_t vz = fdiv(row, P.d_xh);
      const int vy = row - vz * P.XH;
      const int gz = iz0 + (int)vz, gy = iy0 + vy, gx = ix0 + vx;
      const int c = (ci_t0 + pl) * 16 + half * 8;
      float v[8];
      if ((unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi && c < a.CPi) {
        Store<T>::ld8(xg + ((((size_t)b * a.Di + gz) * a.Hi + gy) * a.Wi + gx) * a.CPi + c, v);
        if (a.in_scale) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a.in_scale[c + j], a.in_shift[c + j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      put_chunk<NP, T>(xt + pl * P.xplane + vox * WG_VSB + half * 16, P.lo_off, v);
    }
    __syncthreads();

    // ---- MFMA: every row of the tile is one K step of 32 voxels -------------------------------------
    const int nrows = P.TZ * P.TY;
    for (int row = 0; row < nrows; ++row) {
      const int rz = row / P.TY, ry = row - rz * P.TY;
      const int arow = row * 32 * WG_VSB;
      const int brow = ((rz * a.sD * P.XH + ry * a.sH) * P.XW) * WG_VSB;
      bf16x8 af[COB], afl[COB];
#pragma unroll
      for (int c = 0; c < COB; ++c) {
        const unsigned char* base = dzt + c * P.dzplane + arow;
        af[c] = tr_read2(base + a_off0, base + a_off1);
        if (NP == 2) afl[c] = tr_read2(base + P.lo_off + a_off0, base + P.lo_off + a_off1);
      }
#pragma unroll
      for (int tt = 0; tt < WG_TAPS_PER_WAVE; ++tt) {
        if (tt < ntw) {
#pragma unroll
          for (int i = 0; i < CIB; ++i) {
            const unsigned char* base = xt + i * P.xplane + brow + tapoff[tt];
            const bf16x8 bf = tr_read2(base + b_off0, base + b_off1);
            bf16x8 bfl;
            if (NP == 2) bfl = tr_read2(base + P.lo_off + b_off0, base + P.lo_off + b_off1);
#pragma unroll
            for (int c = 0; c < COB; ++c) {
              acc[tt][c][i] = SP_MFMA16(af[c], bf, acc[tt][c][i], 0, 0, 0);
              if (NP == 2) {
                acc[tt][c][i] = SP_MFMA16(af[c], bfl, acc[tt][c][i], 0, 0, 0);
                acc[tt][c][i] = SP_MFMA16(afl[c], bf, acc[tt][c][i], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }

  // ---- flush: D[row = co = lg*4+j][col = ci = li] -> dw_acc[tap][co][ci] -------------------------------
  const int CoP = a.CoT * 16, CiP = a.CiT * 16;
  // parts mode: this workgroup's own block of partial sums (plain stores), else fp32 atomics into the one block
  float* prow = a.dw_acc + (a.parts ? (size_t)blockIdx.x * a.ntap * CoP * CiP : (size_t)0);
#pragma unroll
  for (int tt = 0; tt < WG_TAPS_PER_WAVE; ++tt) {
    if (tt < ntw) {
#pragma unroll
      for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int i = 0; i < CIB; ++i) {
          if (co_t0 + c < a.CoT && ci_t0 + i < a.CiT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int co = (co_t0 + c) * 16 + lg * 4 + j, ci = (ci_t0 + i) * 16 + li;
              float* dst = prow + ((size_t)(tap0 + tt) * CoP + co) * CiP + ci;
              if (a.parts) *dst = acc[tt][c][i][j]; else atomicAdd(dst, acc[tt][c][i][j]);
            }
          }
        }
    }
  }
}

template <int NP, typename T>
static int wgrad_dispatch(const WgradDev& P, int COB, int CIB, dim3 grid, int lds_bytes, hipStream_t st) {
#define WG_CASE(C_, I_)                                                                              \
  if (COB == C_ && CIB == I_) {                                                                      \
    auto kern = wgrad_kernel<NP, C_, I_, T>;                                                         \
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_wgrad");                                               \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, P);                                     \
    SP_CHECK_LAUNCH("sp_conv3d_wgrad");                                                              \
    return SP_OK;                                                                                    \
  }
  WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(4, 1)
#undef WG_CASE
  sp_set_error("sp_conv3d_wgrad: no kernel for COB=%d CIB=%d", COB, CIB);
  return SP_EINVAL;
}

int sp_conv3d_wgrad_dma(const sp_wgrad_args* a, sp_stream_t stream);   // sp_wgrad_dma.hip
int sp_wgrad_pw_try(const sp_wgrad_args* a, hipStream_t st);            // sp_wgrad_pw.hip

extern "C" int sp_conv3d_wgrad(const sp_wgrad_args* a, sp_stream_t stream) {
  SP_CHECK_ARG(a && a->x && a->dz && a->dw_acc && a->taps, "sp_conv3d_wgrad: null pointer");
  SP_CHECK_ARG(a->CPi % 8 == 0 && a->CPo % 8 == 0, "sp_conv3d_wgrad: channel pitch must be a multiple of 8");
  SP_CHECK_ARG(a->ntap >= 1 && a->ntap <= 4 * WG_TAPS_PER_WAVE, "sp_conv3d_wgrad: ntap=%d out of range", a->ntap);
  SP_CHECK_ARG(a->CoT * 16 >= a->CPo && a->CiT * 16 >= a->CPi, "sp_conv3d_wgrad: tile counts too small");
  SP_CHECK_ARG(a->nblocks >= 1, "sp_conv3d_wgrad: nblocks");
  {     // pointwise layers with per-workgroup partial blocks: the streaming kernel of sp_wgrad_pw.hip
    const int rc = sp_wgrad_pw_try(a, reinterpret_cast<hipStream_t>(stream));
    if (rc <= 0) return rc;
  }
  if (a->dma) return sp_conv3d_wgrad_dma(a, stream);
  WgradDev P;
  P.a = *a;
  // block tile blocking of (cout, cin) tiles: COB*CIB <= 6 accumulator tiles per tap
  int COB = a->CoT >= 4 ? 4 : (a->CoT >= 2 ? 2 : 1);
  int CIB = a->CiT >= 3 ? 3 : a->CiT;
  if (COB == 4) CIB = 1;
  if (a->cib > 0 && a->cib <= CIB) CIB = a->cib;          // caller's blocking of the cin tiles (grid.z grows accordingly)
  P.mdz = P.mdy = P.mdx = 0;
  const int kD = a->kD, kH = a->kH, kW = a->kW;
  SP_CHECK_ARG(kD >= 1 && kH >= 1 && kW >= 1 && kD <= 8 && kH <= 8 && kW <= 8, "sp_conv3d_wgrad: tap extent");
  // tile: TY x TZ rows of 32 voxels; sized so that the staged planes fit LDS
  const int np = a->dtype == SP_F32 ? 2 : 1;
  int TZ = 4, TY = 2;
  for (;;) {
    P.TZ = TZ; P.TY = TY;
    P.XD = (TZ - 1) * a->sD + kD; P.XH = (TY - 1) * a->sH + kH; P.XW = 31 * a->sW + kW;
    P.xplane = (P.XD * P.XH * P.XW * WG_VSB + 64 + 15) & ~15;
    P.dzplane = TZ * TY * 32 * WG_VSB;
    P.lo_off = CIB * P.xplane + COB * P.dzplane;
    const long need = (long)P.lo_off * np;
    if (need <= 150 * 1024 || (TZ == 1 && TY == 1)) break;
    if (TZ > 1) TZ >>= 1; else TY >>= 1;
  }
  const int lds_bytes = P.lo_off * np;
  SP_CHECK_ARG(lds_bytes <= 160 * 1024, "sp_conv3d_wgrad: tile does not fit LDS (%d bytes)", lds_bytes);
  P.ntx = (a->Wo + 31) / 32; P.nty = (a->Ho + P.TY - 1) / P.TY; P.ntz = (a->Do + P.TZ - 1) / P.TZ;
  P.d_tx = make_fastdiv(P.ntx); P.d_ty = make_fastdiv(P.nty); P.d_tz = make_fastdiv(P.ntz);
  P.d_xw = make_fastdiv(P.XW); P.d_xh = make_fastdiv(P.XH);
  const uint64_t nt = (uint64_t)P.ntx * P.nty * P.ntz * a->B;
  SP_CHECK_ARG(nt < (1ull << 31), "sp_conv3d_wgrad: too many tiles");
  P.ntiles = (uint32_t)nt;
  uint32_t gx = (a->parts || a->nblocks < (int64_t)nt) ? a->nblocks : (uint32_t)nt;   // parts: every block is written
  dim3 grid(gx, (a->CoT + COB - 1) / COB, (a->CiT + CIB - 1) / CIB);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == SP_BF16) return wgrad_dispatch<1, bf16_t>(P, COB, CIB, grid, lds_bytes, st);
  if (a->dtype == SP_F32) return wgrad_dispatch<2, float>(P, COB, CIB, grid, lds_bytes, st);
  sp_set_error("sp_conv3d_wgrad: bad dtype %d", a->dtype);
  return SP_EINVAL;
}

// sum over the nparts partial blocks of one accumulator entry: 32 consecutive entries x 8 row lanes per workgroup
// (128-byte row segments), LDS-reduced; lane group 0 continues with the total
__device__ __forceinline__ bool wgrad_part_sum(const float* __restrict__ acc, int nparts, int64_t total, int64_t& idx,
                                               float& v) {
  __shared__ float red[8][33];
  const int el = threadIdx.x & 31, rl = threadIdx.x >> 5;
  idx = (int64_t)blockIdx.x * 32 + el;
  float s = 0.f;
  if (idx < total) {
    int r = rl;
    for (; r + 56 < nparts; r += 64) {      // eight independent loads per trip (four: 17 us per launch at 512 blocks, latency-bound)
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = acc[(size_t)(r + 8 * k) * total + idx];
      s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; r + 24 < nparts; r += 32) {
      const float a0 = acc[(size_t)r * total + idx], a1 = acc[(size_t)(r + 8) * total + idx];
      const float a2 = acc[(size_t)(r + 16) * total + idx], a3 = acc[(size_t)(r + 24) * total + idx];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; r < nparts; r += 8) s += acc[(size_t)r * total + idx];
  }
  red[rl][el] = s;
  __syncthreads();
  if (rl != 0 || idx >= total) return false;
  v = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) v += red[i][el];
  return true;
}

__global__ __launch_bounds__(256) void wgrad_finish_parts_kernel(const float* __restrict__ acc, int nparts,
                                                                 const int32_t* __restrict__ tapsrc, int ntap, int CoP,
                                                                 int CiP, int Cout, int Cin, int64_t sCo, int64_t sCi,
                                                                 float* __restrict__ dw, const double* __restrict__ dbias_sums,
                                                                 float* __restrict__ dbias_grad, int nbias, int dbs) {
  const int64_t total = (int64_t)ntap * CoP * CiP;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (dbias_grad && gid < nbias) dbias_grad[gid] += (float)sp_rows_sum(dbias_sums, (int)gid, dbs);
  int64_t idx;
  float v;
  if (!wgrad_part_sum(acc, nparts, total, idx, v)) return;
  const int ci = idx % CiP;
  const int co = (idx / CiP) % CoP;
  const int t = idx / ((int64_t)CiP * CoP);
  if (co < Cout && ci < Cin) dw[co * sCo + ci * sCi + tapsrc[t]] += v;
}

__global__ void wgrad_finish_kernel(float* __restrict__ acc, const int32_t* __restrict__ tapsrc, int ntap,
                                    int CoP, int CiP, int Cout, int Cin, int64_t sCo, int64_t sCi,
                                    float* __restrict__ dw, const double* __restrict__ dbias_sums,
                                    float* __restrict__ dbias_grad, int nbias, int dbs) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)ntap * CoP * CiP;       // every accumulator entry is visited (and cleared)
  if (dbias_grad && idx < nbias) dbias_grad[idx] += (float)sp_rows_sum(dbias_sums, (int)idx, dbs);
  if (idx >= total) return;
  const int ci = idx % CiP;
  const int co = (idx / CiP) % CoP;
  const int t = idx / ((int64_t)CiP * CoP);
  const float v = acc[idx];
  acc[idx] = 0.f;                                        // ready for the next step: no separate memset
  if (co < Cout && ci < Cin) dw[co * sCo + ci * sCi + tapsrc[t]] += v;
}

extern "C" int sp_wgrad_finish(float* dw_acc, int32_t nparts, const int32_t* tapsrc, int32_t ntap, int32_t CoP, int32_t CiP,
                               int32_t Cout, int32_t Cin, int64_t sCo, int64_t sCi, float* dw,
                               const double* dbias_sums, float* dbias_grad, int32_t nbias, int32_t dbias_stride, sp_stream_t stream) {
  SP_CHECK_ARG(dw_acc && tapsrc && dw && Cout <= CoP && Cin <= CiP && nparts >= 1, "sp_wgrad_finish: bad arguments");
  SP_CHECK_ARG(!dbias_grad || (dbias_sums && nbias <= ntap * CoP * CiP), "sp_wgrad_finish: bias arguments");
  const int64_t total = (int64_t)ntap * CoP * CiP;
  if (nparts > 1) {
    SP_CHECK_ARG(!dbias_grad || nbias <= (total + 31) / 32 * 256, "sp_wgrad_finish: bias arguments");
    hipLaunchKernelGGL(wgrad_finish_parts_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), dw_acc, nparts, tapsrc, ntap, CoP, CiP, Cout, Cin, sCo, sCi,
                       dw, dbias_sums, dbias_grad, nbias, dbias_stride);
    SP_CHECK_LAUNCH("sp_wgrad_finish");
    return SP_OK;
  }
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dw_acc, tapsrc, ntap, CoP, CiP, Cout, Cin, sCo, sCi, dw,
                     dbias_sums, dbias_grad, nbias, dbias_stride);
  SP_CHECK_LAUNCH("sp_wgrad_finish");
  return SP_OK;
}
