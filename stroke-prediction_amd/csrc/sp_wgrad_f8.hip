// fp8 weight gradient of the stride-1 un-padded 3x3x3 convolutions (Unet3D.py:19,22 inside the 4-scale topology
// Unet3D.py:95-146; BASELINE.json configs[4] "fp8 MFMA"): dw[tap][co][ci] = sum over output voxels of dz[v][co] * x[v + tap][ci]
// on v_mfma_f32_16x16x128_f8f6f4 with dz in e5m2 (A operand, scaled by a power of two when it was quantised) and x in e4m3
// (B operand), both read from the PLANE-MAJOR fp8 tensors the fp8 forward / data-gradient convolutions already use
// ([C/16][B][D][H][W][16 bytes], csrc/sp_conv_zm8.hip).
//
// Same march as the bf16 row-sliding kernel (csrc/sp_wgrad_zr.hip; read that header first): a workgroup of 4 waves owns a
// column of TY output rows x 32 voxels and a 32 x 32 channel block (one (cout tile, cin tile) pair per wave, 27 accumulator
// tiles each), marches over the input planes with rings of three x and three dz slots filled by LDS-DMA two planes ahead, and
// keeps the dz fragments of a plane in registers for the three steps in which it is the dz = 0, 1, 2 neighbour of the marching
// input plane.  What one-byte operands change:
//
//   * K = 128 per MFMA = FOUR output rows x 32 voxels: lane group kb = lane >> 4 of the instruction takes row 4 rg + kb, its 32
//     bytes are the 32 voxels of that row; both operands use the same map, which is all the instruction asks for;
//   * a lane needs 32 consecutive VOXELS of ONE channel while memory holds 16 channels per voxel: ds_read_b64_tr_b8 does the
//     transpose -- 16 lanes with consecutive 8-byte addresses (8 voxels x 16 channels) get back lane i = channel i, 8 voxels
//     (measured lane map: tools/probes/probe_f8.hip, gpurun_out/r3_probe_f8.txt); four such reads make a fragment;
//   * tile rows are padded to 40 voxels (640 bytes = 128 mod 256): the two lane groups of a half-wave then read disjoint banks;
//   * the input fragment of rows (4 rg + dy ... + 3) at x shift dx serves the three dz taps: 9 fragment reads per 27 MFMAs.
//
// Results: per-workgroup partial blocks [27][CoP][CiP] fp32 (sp_wgrad_finish_folded sums them; the accumulators carry the
// scale of dz, undone there through `acc_scale`).
#include <stdlib.h>
#include "sp_common.h"

#define F8W_XW 40                    // row pitch of both tiles in voxels (34 / 32 of them real)
#define F8W_VB 16                    // bytes per voxel of a 16-channel fp8 plane

typedef int f8w_x32 __attribute__((ext_vector_type(8)));
typedef int f8w_i2 __attribute__((ext_vector_type(2)));

__device__ uint4 sp_f8w_zero_page[64];

struct WgradF8Dev {
  sp_wgrad_f8_args a;
  int32_t nty, ntx, xcd;
  uint32_t ncols;
  FastDiv d_tx, d_ty;
};

#define F8W_SYNC(N)                                                  \
  do {                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");         \
    __builtin_amdgcn_s_barrier();                                    \
    asm volatile("" ::: "memory");                                   \
  } while (0)

// one K = 128 fragment: rows (r0 + kb) of a tile plane, voxels x0 .. x0 + 31, this lane's channel (lane & 15)
__device__ __forceinline__ f8w_x32 f8w_frag(const unsigned char* p) {
  typedef __attribute__((address_space(3))) f8w_i2 lds_i2;
  const f8w_i2 r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i2*)(p));
  const f8w_i2 r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i2*)(p + 8 * F8W_VB));
  const f8w_i2 r2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i2*)(p + 16 * F8W_VB));
  const f8w_i2 r3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i2*)(p + 24 * F8W_VB));
  return f8w_x32{r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};
}

template <int TYG>
struct F8wCfg {
  static constexpr int TY = 4 * TYG, XH = TY + 2;
  static constexpr int XT = XH * F8W_XW * F8W_VB;                 // bytes of one 16-channel x plane tile
  static constexpr int DT = TY * F8W_XW * F8W_VB;                 // ... of one dz plane tile
  static constexpr int NXC = 2 * XH * F8W_XW, NJX = (NXC + 255) / 256, XSB = NJX * 4096;
  static constexpr int NDC = 2 * TY * F8W_XW, NJD = (NDC + 255) / 256, DSB = NJD * 4096;
  static constexpr int NS = 3, D = 2;
  static constexpr int LDS = NS * (XSB + DSB);
};

template <int TYG>
__global__ __launch_bounds__(256, 2) void wgrad_f8_kernel(const WgradF8Dev P) {
  typedef F8wCfg<TYG> C;
  constexpr int NJX = C::NJX, NJD = C::NJD, NS = C::NS, D = C::D, NJ = NJX + NJD;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_wgrad_f8_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = lane >> 4, li = lane & 15, lg = lane >> 4;
  const int co_t0 = blockIdx.y * 2, ci_t0 = blockIdx.z * 2;
  const int wco = wave >> 1, wci = wave & 1;                       // this wave's (cout tile, cin tile) of the 2 x 2 block
  unsigned char* xring = lds;
  unsigned char* dring = lds + NS * C::XSB;
  const unsigned char* zeros = reinterpret_cast<const unsigned char*>(sp_f8w_zero_page) + lane * 16;

  // ---- per-lane DMA plans: chunk c = (wave + 4 j) * 64 + lane -> (plane, tile row, tile voxel)
  uint32_t relx[NJX], reld[NJD];
  int crdx[NJX], crdd[NJD];
#pragma unroll
  for (int j = 0; j < NJX; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;
    const int pl = c / (C::XH * F8W_XW), vox = c - pl * (C::XH * F8W_XW);
    const int vy = vox / F8W_XW, vx = vox - vy * F8W_XW;
    const bool ok = c < C::NXC && vx < 34 && ci_t0 + pl < a.CiT;
    relx[j] = (uint32_t)(ci_t0 + pl) * (uint32_t)a.x_plane + (uint32_t)((vy * a.Wi + vx) * F8W_VB);
    crdx[j] = ok ? (vy | (vx << 8)) : -1;
  }
#pragma unroll
  for (int j = 0; j < NJD; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;
    const int pl = c / (C::TY * F8W_XW), vox = c - pl * (C::TY * F8W_XW);
    const int ry = vox / F8W_XW, rx = vox - ry * F8W_XW;
    const bool ok = c < C::NDC && rx < 32 && co_t0 + pl < a.CoT;
    reld[j] = (uint32_t)(co_t0 + pl) * (uint32_t)a.dz_plane + (uint32_t)((ry * a.Wo + rx) * F8W_VB);
    crdd[j] = ok ? (ry | (rx << 8)) : -1;
  }
  // lane offset of a fragment read: row kb of the 4-row group, this lane's 8 bytes of the 16-lane transpose group
  const int loff = kb * (F8W_XW * F8W_VB) + li * 8;
  const int aoff = wco * C::DT + loff;
  const int boff = wci * C::XT + loff;

  f32x4 acc[3][3][3];
#pragma unroll
  for (int i = 0; i < 27; ++i) (&acc[0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f8w_x32 zfrag = {0, 0, 0, 0, 0, 0, 0, 0};

  const uint32_t xplane_b = (uint32_t)a.Hi * a.Wi * F8W_VB, dplane_b = (uint32_t)a.Ho * a.Wo * F8W_VB;
  const uint32_t vb = P.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const uint64_t T = (uint64_t)P.ncols * a.Do;
  uint64_t pos = T * vb / gridDim.x;
  const uint64_t pend = T * (vb + 1) / gridDim.x;
  while (pos < pend) {
    const uint32_t col = (uint32_t)(pos / (uint32_t)a.Do);
    const int z0 = (int)(pos - (uint64_t)col * a.Do);
    const int z1 = (int)min((uint64_t)a.Do, (uint64_t)z0 + (pend - pos));
    pos += (uint64_t)(z1 - z0);
    const int np = z1 - z0, nsteps = np + 2;
    uint32_t t = col;
    uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
    q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; const int b = q;
    const int oy0 = ty * C::TY, ox0 = tx * 32;
    const unsigned char* xcol = reinterpret_cast<const unsigned char*>(a.x) +
        ((((int64_t)b * a.Di + z0) * a.Hi + oy0) * a.Wi + ox0) * F8W_VB;
    const unsigned char* dcol = reinterpret_cast<const unsigned char*>(a.dz) +
        ((((int64_t)b * a.Do + z0) * a.Ho + oy0) * a.Wo + ox0) * F8W_VB;
    const unsigned char* srcx[NJX]; uint32_t strx[NJX];
    const unsigned char* srcd[NJD]; uint32_t strd[NJD];
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int vy = crdx[j] & 0xff, vx = (crdx[j] >> 8) & 0xff;
      const bool ok = crdx[j] >= 0 && oy0 + vy < a.Hi && ox0 + vx < a.Wi;
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
    // issue1(j, k, slot): one DMA of step k (input plane z0 + k, dz plane z0 + k).  Past the piece's last plane the previous
    // plane is fetched again (never read): every step issues the same NJ DMAs, which keeps the waits countable.
    auto issue1 = [&](int j, int k, int slot) {
      if (j < NJX) {
        sp_dma16_nc(srcx[j < NJX ? j : 0], xring + slot * C::XSB + wave * 1024 + j * 4096);
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
    f8w_x32 A[3][TYG];
#pragma unroll
    for (int r = 0; r < TYG; ++r) A[1][r] = A[2][r] = zfrag;
    F8W_SYNC(0);
    issue(0, 0); issue(1, 1);
    int sl = 0;
    for (int s = 0; s < nsteps; ++s) {
      F8W_SYNC((D - 1) * NJ);
      const int sn = sl == 0 ? NS - 1 : sl - 1;
      const unsigned char* ap = dring + sl * C::DSB + aoff;
      const unsigned char* bp = xring + sl * C::XSB + boff;
      if (s < np) {
#pragma unroll
        for (int r = 0; r < TYG; ++r) A[0][r] = f8w_frag(ap + r * 4 * (F8W_XW * F8W_VB));
      } else {
#pragma unroll
        for (int r = 0; r < TYG; ++r) A[0][r] = zfrag;
      }
      constexpr int NG = 3 * TYG;                           // MFMA groups of a step: (row group, dy)
      f8w_x32 bq[2][3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) bq[0][dx] = f8w_frag(bp + dx * F8W_VB);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int rg = gi / 3, dy = gi - rg * 3;
        if (gi + 1 < NG) {
          const int rg1 = (gi + 1) / 3, dy1 = (gi + 1) - rg1 * 3;
          const unsigned char* brow = bp + (rg1 * 4 + dy1) * (F8W_XW * F8W_VB);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) bq[(gi + 1) & 1][dx] = f8w_frag(brow + dx * F8W_VB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (j * NG / NJ == gi) issue1(j, s + D, sn);
#pragma unroll
        for (int dzz = 0; dzz < 3; ++dzz)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            acc[dzz][dy][dx] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[dzz][rg], bq[gi & 1][dx], acc[dzz][dy][dx], 1, 0, 0, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < TYG; ++r) { A[2][r] = A[1][r]; A[1][r] = A[0][r]; }
      sl = sl == NS - 1 ? 0 : sl + 1;
    }
  }
  // ---- flush: this workgroup's block of partial sums
  F8W_SYNC(0);
  const int CoP = a.CoT * 16, CiP = a.CiT * 16;
  float* prow = a.dw_acc + (size_t)blockIdx.x * 27 * CoP * CiP;
  if (co_t0 + wco < a.CoT && ci_t0 + wci < a.CiT) {
#pragma unroll
    for (int tp = 0; tp < 27; ++tp)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = (co_t0 + wco) * 16 + lg * 4 + j, ci = (ci_t0 + wci) * 16 + li;
        prow[((size_t)tp * CoP + co) * CiP + ci] = (&acc[0][0][0])[tp][j];
      }
  }
}

extern "C" int sp_conv3d_wgrad_f8(const sp_wgrad_f8_args* a, sp_stream_t stream) {
  SP_CHECK_ARG(a && a->x && a->dz && a->dw_acc, "sp_conv3d_wgrad_f8: null pointer");
  SP_CHECK_ARG(a->B > 0 && a->Do > 0 && a->Ho > 0 && a->Wo > 0 && a->Di == a->Do + 2 && a->Hi == a->Ho + 2 && a->Wi == a->Wo + 2,
               "sp_conv3d_wgrad_f8: stride-1 un-padded 3x3x3 geometry only");
  SP_CHECK_ARG(a->CoT >= 2 && a->CiT >= 2 && a->CoT % 2 == 0 && a->CiT % 2 == 0, "sp_conv3d_wgrad_f8: whole 32 x 32 channel blocks (CoT %d, CiT %d)", a->CoT, a->CiT);
  SP_CHECK_ARG(a->nblocks >= 1 && (uint64_t)a->x_plane * a->CiT < (1ull << 32) && (uint64_t)a->dz_plane * a->CoT < (1ull << 32) &&
               a->x_plane >= (int64_t)a->B * a->Di * a->Hi * a->Wi * 16 && a->dz_plane >= (int64_t)a->B * a->Do * a->Ho * a->Wo * 16,
               "sp_conv3d_wgrad_f8: plane sizes / 32-bit offsets");
  SP_CHECK_ARG(a->Ho <= 255 * 8, "sp_conv3d_wgrad_f8: too many rows");
  WgradF8Dev P;
  P.a = *a;
  constexpr int TYG = 1;
  P.nty = (a->Ho + 4 * TYG - 1) / (4 * TYG); P.ntx = (a->Wo + 31) / 32;
  P.ncols = (uint32_t)(a->B * P.nty * P.ntx);
  P.d_tx = make_fastdiv(P.ntx); P.d_ty = make_fastdiv(P.nty);
  const uint32_t gx = a->nblocks;
  P.xcd = (gx % 8 == 0 && gx >= 8) ? 1 : 0;
  dim3 grid(gx, a->CoT / 2, a->CiT / 2);
  auto kern = wgrad_f8_kernel<TYG>;
  const int lds_bytes = F8wCfg<TYG>::LDS;
  SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_wgrad_f8");
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, reinterpret_cast<hipStream_t>(stream), P);
  SP_CHECK_LAUNCH("sp_conv3d_wgrad_f8");
  return SP_OK;
}
