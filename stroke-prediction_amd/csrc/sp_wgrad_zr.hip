// Weight gradient of the stride-1 3x3x3 convolutions, padding 0..2 (every Block3x3x3 conv of the U-Net, Unet3D.py:19,22; the
// padded layers of the CAE, Cae3D.py:41-70,186-218, on their materialised BatchNorm output),
// bf16, "row-sliding" z-marching variant.  Same GEMM view as sp_wgrad_dma.hip (dw[tap][co][ci] = sum over output voxels
// of dz[v][co] * x[v + tap][ci], K = 32 voxels along x per MFMA, both operands through LDS and ds_read_b64_tr_b16), but
// organised around the INPUT row instead of the tap, which is what the LDS bandwidth asks for:
//
//   the fragment of input row (zi, yi) at x shift dx is the B operand of NINE taps -- (dz, dy, dx) for the nine output
//   rows (zi - dz, yi - dy) -- so it is read from LDS once and multiplied with nine dz fragments that are already in
//   registers: the dz rows of a plane stay in registers for the three steps in which the plane is the dz = 0, 1, 2
//   neighbour of the marching input plane.  Per input row a wave reads 3 B fragments and (once per plane) RW A fragments
//   for 27 MFMAs per output row: 0.22-0.33 fragment reads per MFMA, against 1.29 in the tap-major kernels whose
//   16x16-tile layers were LDS-bound at ~25 % MFMA utilisation (profiles/r02_wgrad_pmc.txt).
//
//   A workgroup (4 waves, two per CU) owns a column of TY output rows x 32 voxels and marches over input planes; per
//   step ONE x plane and ONE dz plane are DMA'd (global_load_lds_dwordx4) into rings of three slots, two steps ahead,
//   with a counted s_waitcnt and one barrier per step.  Border chunks are DMA'd from a zero page, so every wave issues
//   the same number of DMAs per step and the count is a compile-time constant.
//   Waves split the (cout tile, cin tile) pairs of the workgroup's channel block first, then the tile rows; the partial
//   sums of waves that share a pair are added through LDS before the flush.
//
// Results land in the same per-workgroup partial blocks as the other DMA kernels (parts mode, summed by
// sp_wgrad_finish_folded).
#include <stdlib.h>
#include "sp_common.h"

#define ZR_VSB 32                    // bytes per voxel in LDS (16 bf16 channels)
#define ZR_XW 34                     // input voxels per row of a 32-voxel output tile

__device__ uint4 sp_zr_zero_page[64];   // source of border chunks (zero-initialised device memory)

struct WgradZrDev {
  sp_wgrad_args a;
  int32_t nty, ntx, xcd;
  uint32_t ncols;
  FastDiv d_tx, d_ty;
};

__device__ __forceinline__ bf16x8 zr_tr_read2(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p0));
  bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p1));
  return __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
}

#define ZR_SYNC(N)                                                   \
  do {                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");         \
    __builtin_amdgcn_s_barrier();                                    \
    asm volatile("" ::: "memory");                                   \
  } while (0)

template <int COB, int CIB, int RW>
struct ZrCfg {
  static constexpr int PAIRS = COB * CIB, RS = 4 / PAIRS, TY = RS * RW, XH = TY + 2;
  static constexpr int XT = XH * ZR_XW * ZR_VSB;                 // bytes of one 16-channel input plane tile
  static constexpr int DT = TY * 32 * ZR_VSB;                    // bytes of one 16-channel dz plane tile
  static constexpr int NXC = CIB * XH * ZR_XW * 2, NJX = (NXC + 255) / 256, XSB = NJX * 4096;
  static constexpr int NDC = COB * TY * 32 * 2, NJD = (NDC + 255) / 256, DSB = NJD * 4096;
  static constexpr int NS = 3, D = 2;
  static constexpr int RING = NS * (XSB + DSB);
  static constexpr int RED = RS > 1 ? PAIRS * 27 * 1024 : 0;
  static constexpr int LDS = RING > RED ? RING : RED;
};

template <int COB, int CIB, int RW>
__global__ __launch_bounds__(256, 2) void wgrad_zr_kernel(const WgradZrDev P) {
  typedef ZrCfg<COB, CIB, RW> C;
  constexpr int NJX = C::NJX, NJD = C::NJD, NS = C::NS, D = C::D, NJ = NJX + NJD;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int co_t0 = blockIdx.y * COB, ci_t0 = blockIdx.z * CIB;
  const int pair = wave / C::RS, rbase = (wave % C::RS) * RW;
  const int wco = pair / CIB, wci = pair % CIB;
  unsigned char* xring = lds;
  unsigned char* dring = lds + NS * C::XSB;
  const int xpitch = a.x_plane ? 16 : a.CPi;
  const unsigned char* zeros = reinterpret_cast<const unsigned char*>(sp_zr_zero_page) + lane * 16;

  // ---- per-lane DMA plans (column-invariant): byte offset inside the column's plane, tile coordinates for the border test
  uint32_t relx[NJX], reld[NJD];
  int crdx[NJX], crdd[NJD];
#pragma unroll
  for (int j = 0; j < NJX; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;
    const int half = c & 1, rest = c >> 1;
    const int pl = rest / (C::XH * ZR_XW), vox = rest - pl * (C::XH * ZR_XW);
    const int vy = vox / ZR_XW, vx = vox - vy * ZR_XW;
    const bool ok = c < C::NXC && ci_t0 + pl < a.CiT;
    relx[j] = a.x_plane ? (uint32_t)((vy * a.Wi + vx) * 32 + half * 16) + (uint32_t)(ci_t0 + pl) * (uint32_t)(a.x_plane * 2)
                        : (uint32_t)(((vy * a.Wi + vx) * xpitch + (ci_t0 + pl) * 16 + half * 8) * 2);
    crdx[j] = ok ? (vy | (vx << 8)) : -1;
  }
#pragma unroll
  for (int j = 0; j < NJD; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;
    const int half = c & 1, rest = c >> 1;
    const int pl = rest / (C::TY * 32), vox = rest - pl * (C::TY * 32);
    const int ry = vox >> 5, rx = vox & 31;
    const bool ok = c < C::NDC && co_t0 + pl < a.CoT;
    reld[j] = (uint32_t)(((ry * a.Wo + rx) * a.CPo + (co_t0 + pl) * 16 + half * 8) * 2);
    crdd[j] = ok ? (ry | (rx << 8)) : -1;
  }
  // transposed-read lane offsets (same voxel permutation for both operands; see sp_wgrad_dma.hip)
  const int vq0 = ((lg & 1) ? 2 * lg + 1 : 2 * lg) * 4 + lq;
  const int vq1 = ((lg & 1) ? 2 * lg : 2 * lg + 1) * 4 + lq;
  const int off0 = vq0 * ZR_VSB + lp * 8, off1 = vq1 * ZR_VSB + lp * 8;
  const int aoff = wco * C::DT + rbase * 32 * ZR_VSB;             // this wave's first dz row inside a slot
  const int boff = wci * C::XT + rbase * ZR_XW * ZR_VSB;          // this wave's first input row inside a slot

  f32x4 acc[3][3][3];
#pragma unroll
  for (int i = 0; i < 27; ++i) (&acc[0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 zfrag = {0, 0, 0, 0, 0, 0, 0, 0};

  const uint32_t xplane_b = (uint32_t)a.Hi * a.Wi * xpitch * 2, dplane_b = (uint32_t)a.Ho * a.Wo * a.CPo * 2;
  const uint32_t vb = P.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const uint64_t T = (uint64_t)P.ncols * a.Do;
  uint64_t pos = T * vb / gridDim.x;
  const uint64_t pend = T * (vb + 1) / gridDim.x;
  while (pos < pend) {
    const uint32_t col = (uint32_t)(pos / (uint32_t)a.Do);
    const int z0 = (int)(pos - (uint64_t)col * a.Do);
    const int z1 = (int)min((uint64_t)a.Do, (uint64_t)z0 + (pend - pos));
    pos += (uint64_t)(z1 - z0);
    const int np = z1 - z0, nsteps = np + 2;                       // output planes of this piece, input planes it needs
    uint32_t t = col;
    uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
    q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; const int b = q;
    const int oy0 = ty * C::TY, ox0 = tx * 32;
    // padded layers (o0 = -padding <= 0): the input window starts o0 voxels before the output tile; planes, rows and columns
    // outside the input come from the zero page.  xcol may point in front of the tensor: it is only dereferenced for
    // chunks inside it.
    const int iy0 = oy0 + a.o0H, ix0 = ox0 + a.o0W, izs = z0 + a.o0D;
    const unsigned char* xcol = reinterpret_cast<const unsigned char*>(a.x) +
        ((((int64_t)b * a.Di + izs) * a.Hi + iy0) * a.Wi + ix0) * xpitch * 2;
    const unsigned char* dcol = reinterpret_cast<const unsigned char*>(a.dz) +
        ((((int64_t)b * a.Do + z0) * a.Ho + oy0) * a.Wo + ox0) * a.CPo * 2;
    const unsigned char* srcx[NJX]; uint32_t strx[NJX];
    const unsigned char* srcd[NJD]; uint32_t strd[NJD];
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int vy = crdx[j] & 0xff, vx = (crdx[j] >> 8) & 0xff;
      const bool ok = crdx[j] >= 0 && (unsigned)(iy0 + vy) < (unsigned)a.Hi && (unsigned)(ix0 + vx) < (unsigned)a.Wi;
      srcx[j] = ok ? xcol + relx[j] : zeros;
      strx[j] = ok ? xplane_b : 0u;
    }
#pragma unroll
    for (int j = 0; j < NJD; ++j) {
      const int ry = crdd[j] & 0xff, rx = (crdd[j] >> 8) & 0xff;
      const bool ok = crdd[j] >= 0 && oy0 + ry < a.Ho && ox0 + rx < a.Wo;
      srcd[j] = ok ? dcol + reld[j] : zeros;
      strd[j] = ok ? dplane_b : 0u;
    }
    // issue(k): input plane z0 + k and dz plane z0 + k into slot k % NS.  Past the piece's last plane the previous
    // plane is fetched again (never read): every step issues the same NJ DMAs, which keeps the waits countable.
    // one DMA of issue(k) at a time (j < NJX: input plane, else dz plane), so that the main loop can spread them between
    // the MFMA groups of a step: a DMA costs 60-180 issue cycles that a burst of NJ of them would expose after the barrier
    auto issue1 = [&](int j, int k, int slot) {
      if (j < NJX) {
        const bool zin = (unsigned)(izs + min(k, nsteps - 1)) < (unsigned)a.Di;          // wave-uniform: plane inside the input
        sp_dma16_nc(zin ? srcx[j < NJX ? j : 0] : zeros, xring + slot * C::XSB + wave * 1024 + j * 4096);
        srcx[j < NJX ? j : 0] += (k + 1 < nsteps) ? strx[j < NJX ? j : 0] : 0u;
      } else {
        const int jd = j < NJX ? 0 : j - NJX;
        sp_dma16_nc(srcd[jd], dring + slot * C::DSB + wave * 1024 + jd * 4096);
        srcd[jd] += (k + 1 < np) ? strd[jd] : 0u;
      }
    };
    auto issue = [&](int k, int slot) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) issue1(j, k, slot);
    };
    bf16x8 A[3][RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) A[1][r] = A[2][r] = zfrag;
    ZR_SYNC(0);                                              // the previous column's slots are consumed, its DMAs landed
    issue(0, 0); issue(1, 1);
    int sl = 0;
    for (int s = 0; s < nsteps; ++s) {
      ZR_SYNC((D - 1) * NJ);                                 // step s landed (every wave's share); slot of step s - 1 is free
      const int sn = sl == 0 ? NS - 1 : sl - 1;              // (s + D) % NS == (s - 1) % NS
      const unsigned char* ap = dring + sl * C::DSB + aoff;
      const unsigned char* bp = xring + sl * C::XSB + boff;
      if (s < np) {
#pragma unroll
        for (int r = 0; r < RW; ++r) A[0][r] = zr_tr_read2(ap + r * 32 * ZR_VSB + off0, ap + r * 32 * ZR_VSB + off1);
      } else {
#pragma unroll
        for (int r = 0; r < RW; ++r) A[0][r] = zfrag;
      }
      constexpr int NG = RW + 2;                             // MFMA groups (input rows) of a step
      // B fragments are double-buffered by hand: the reads of row yi + 1 are issued BEFORE the MFMAs of row yi (left
      // alone, the scheduler re-uses one register set and waits for LDS at the head of every group)
      bf16x8 bq[NG][3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) bq[0][dx] = zr_tr_read2(bp + dx * ZR_VSB + off0, bp + dx * ZR_VSB + off1);
#pragma unroll
      for (int yi = 0; yi < NG; ++yi) {
        if (yi + 1 < NG) {
          const unsigned char* brow = bp + (yi + 1) * ZR_XW * ZR_VSB;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) bq[yi + 1][dx] = zr_tr_read2(brow + dx * ZR_VSB + off0, brow + dx * ZR_VSB + off1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // this group's share of the step's DMAs (the slot they fill was released by the barrier above)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (j * NG / NJ == yi) issue1(j, s + D, sn);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int yl = yi - dy;
          if (yl >= 0 && yl < RW) {
#pragma unroll
            for (int dzz = 0; dzz < 3; ++dzz)
#pragma unroll
              for (int dx = 0; dx < 3; ++dx)
                acc[dzz][dy][dx] = SP_MFMA16(A[dzz][yl], bq[yi][dx], acc[dzz][dy][dx], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) { A[2][r] = A[1][r]; A[1][r] = A[0][r]; }
      sl = sl == NS - 1 ? 0 : sl + 1;
    }
  }
  // ---- flush: this workgroup's block of partial sums (waves sharing a channel pair are added through LDS first)
  ZR_SYNC(0);
  const int CoP = a.CoT * 16, CiP = a.CiT * 16;
  float* prow = a.dw_acc + (size_t)vb * 27 * CoP * CiP;      // block vb took piece vb of the (sample-major) march: consecutive blocks = consecutive samples (group-aware finish)
  if (C::RS == 1) {
    if (co_t0 + wco < a.CoT && ci_t0 + wci < a.CiT) {
#pragma unroll
      for (int tp = 0; tp < 27; ++tp)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = (co_t0 + wco) * 16 + lg * 4 + j, ci = (ci_t0 + wci) * 16 + li;
          prow[((size_t)tp * CoP + co) * CiP + ci] = (&acc[0][0][0])[tp][j];
        }
    }
  } else {
    float* red = reinterpret_cast<float*>(lds) + pair * 27 * 256;
#pragma unroll
    for (int rs = 0; rs < C::RS; ++rs) {
      if ((wave % C::RS) == rs) {
#pragma unroll
        for (int tp = 0; tp < 27; ++tp)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float* d = red + tp * 256 + (lg * 4 + j) * 16 + li;
            *d = rs == 0 ? (&acc[0][0][0])[tp][j] : *d + (&acc[0][0][0])[tp][j];
          }
      }
      __syncthreads();
    }
    const float* all = reinterpret_cast<const float*>(lds);
    for (int e = tid; e < C::PAIRS * 27 * 256; e += 256) {
      const int pr = e / (27 * 256), r = e - pr * (27 * 256);
      const int tp = r >> 8, col = (r >> 4) & 15, cil = r & 15;
      const int cot = co_t0 + pr / CIB, cit = ci_t0 + pr % CIB;
      if (cot < a.CoT && cit < a.CiT) prow[((size_t)tp * CoP + cot * 16 + col) * CiP + cit * 16 + cil] = all[e];
    }
  }
}

// returns 1 when the layer is not one this variant handles (the caller falls back to the tap-major kernels)
int sp_wgrad_zr_try(const sp_wgrad_args* a, hipStream_t st) {
  const char* knob = getenv("SP_WGRAD_ZR");          // read per launch: the parity tests compare both kernel families
  if ((knob && atoi(knob) == 0) || !a->parts || a->dtype != SP_BF16 || a->in_scale || a->dz_scale) return 1;
  if (a->kD != 3 || a->kH != 3 || a->kW != 3 || a->ntap != 27) return 1;
  if (a->sD != 1 || a->sH != 1 || a->sW != 1) return 1;
  if (a->o0D > 0 || a->o0H > 0 || a->o0W > 0 || a->o0D < -2 || a->o0H < -2 || a->o0W < -2) return 1;      // padding 0..2
  if (a->Di - 2 * a->o0D != a->Do + 2 || a->Hi - 2 * a->o0H != a->Ho + 2 || a->Wi - 2 * a->o0W != a->Wo + 2) return 1;
  if (a->CPi % 16 || a->CPo % 16 || a->Ho > 255 * 8 || (int64_t)a->Hi * a->Wi * a->CPi * 2 >= (1ll << 31)) return 1;
  if (a->x_plane && (int64_t)a->x_plane * 2 * a->CiT >= (1ll << 32)) return 1;
  const int COB = (a->CoT % 2 == 0) ? 2 : 1, CIB = (a->CiT % 2 == 0) ? 2 : 1;
  WgradZrDev P;
  P.a = *a;
  const int TY = (COB * CIB == 4) ? 4 : (COB * CIB == 2 ? 4 : 8);
  P.nty = (a->Ho + TY - 1) / TY; P.ntx = (a->Wo + 31) / 32;
  P.ncols = (uint32_t)(a->B * P.nty * P.ntx);
  P.d_tx = make_fastdiv(P.ntx); P.d_ty = make_fastdiv(P.nty);
  const uint32_t gx = a->nblocks;
  P.xcd = (gx % 8 == 0 && gx >= 8) ? 1 : 0;
  dim3 grid(gx, a->CoT / COB, a->CiT / CIB);
#define ZR_CASE(C_, I_, R_)                                                                          \
  if (COB == C_ && CIB == I_) {                                                                      \
    static_assert(ZrCfg<C_, I_, R_>::TY == ((C_ * I_ == 1) ? 8 : 4), "tile rows");                   \
    auto kern = wgrad_zr_kernel<C_, I_, R_>;                                                         \
    const int lds_bytes = ZrCfg<C_, I_, R_>::LDS;                                                    \
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_wgrad(zr)");                                           \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, P);                                     \
    SP_CHECK_LAUNCH("sp_conv3d_wgrad(zr)");                                                          \
    return SP_OK;                                                                                    \
  }
  ZR_CASE(1, 1, 2) ZR_CASE(1, 2, 2) ZR_CASE(2, 1, 2) ZR_CASE(2, 2, 4)
#undef ZR_CASE
  return 1;
}
