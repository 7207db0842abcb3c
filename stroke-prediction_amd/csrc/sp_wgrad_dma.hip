// bf16 fast path of the weight gradient for 3x3x3 / 2x2x2 convolutions of stride 1 or 2 with padding 0..2 (every Block3x3x3
// conv of the U-Net, Unet3D.py:19,22; round 4: the CAE's strided and, with swapped roles, transposed layers, Cae3D.py:45-64,
// 178-204 -- the staged input tile is then the dense box the strided taps of a row tile reach, read with a voxel stride).  Same GEMM view as sp_wgrad.hip (K = 32 output voxels along x, both operands
// through LDS and ds_read_b64_tr_b16), restructured after the conv kernel's phase analysis:
//   * persistent workgroups, TWO LDS tile buffers: the LDS-DMA (global_load_lds_dwordx4) of tile t+1 is in
//     flight while tile t feeds the MFMAs; one barrier per tile;
//   * the chunk -> (plane, voxel) decode of every DMA a lane issues is tile-invariant: it is done once before the
//     tile loop (NJ element offsets in registers); per tile a DMA costs one 64-bit add.  Tiles that cross the
//     output border take a masked path (invalid chunks are written as zeros by ds_write);
//   * every wave computes exactly 7 taps (wave 3 repeats tap 26 and drops it at the flush): no guards inside the
//     unrolled MFMA loop (guards make hipcc shuttle accumulators between VGPRs and AGPRs);
//   * the BatchNorm of the input is NOT applied on load (a DMA cannot): dw = scale[ci]*dw_raw + shift[ci]*sum(dz),
//     exact for padding 0, applied by sp_wgrad_finish_folded.
#include <stdlib.h>
#include "sp_common.h"

#ifdef SP_CONV_STAMPS
extern __device__ unsigned long long sp_stamp_buf[32768][6];
#define WSTAMP(var)                                                                           \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#endif
#define WD_TW 7
#define WD_NJX 12
#define WD_NJD 4
#define WD_VSB 32

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct WgradDmaDev {
  sp_wgrad_args a;
  FastDiv d_tx, d_ty, d_tz, d_xw, d_xh, d_xv, d_tv, d_ty_rows;
  uint32_t ntx, nty, ntz, ntiles;
  int32_t TZ, TY, XD, XH, XW, XV, TV;
  int32_t nx_chunks, ndz_chunks, njx, njd, buf_bytes, dz_off, xcd;
};

__device__ __forceinline__ bf16x8 wd_tr_read2(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p0));
  bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p1));
  return __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int COB, int CIB>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgradDmaDev P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int co_t0 = blockIdx.y * COB, ci_t0 = blockIdx.z * CIB;
  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(a.x);
  const bf16_t* __restrict__ dzg = reinterpret_cast<const bf16_t*>(a.dz);

  // zero both buffers once: chunks of channels beyond the pitch are never written by a DMA
  for (int i = tid; i < (2 * P.buf_bytes) / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);

  // ---- tile-invariant DMA plan of this lane.  The x region and the dz region are each padded to whole rounds of
  // 256 chunks, so a DMA instruction never mixes the two tensors: source = wave-uniform tile base (SGPR pair) +
  // per-lane 32-bit byte offset.  Chunks past the end of a region re-load its last chunk (same bytes, harmless).
  uint32_t relx[WD_NJX], reld[WD_NJD];     // byte offsets relative to the tile origin
  int crdx[WD_NJX], crdd[WD_NJD];          // packed tile coordinates (z | y<<8 | x<<16) for the border path
#pragma unroll
  for (int j = 0; j < WD_NJX; ++j) {
    int c = (wave + 4 * j) * 64 + lane;
    c = c < P.nx_chunks ? c : P.nx_chunks - 1;
    const int half = c & 1, rest = c >> 1;
    const uint32_t pl = fdiv(rest, P.d_xv);
    const uint32_t vox = rest - pl * P.XV;
    const uint32_t row = fdiv(vox, P.d_xw);
    const int vx = vox - row * P.XW;
    const uint32_t vz = fdiv(row, P.d_xh);
    const int vy = row - vz * P.XH;
    relx[j] = a.x_plane ? (uint32_t)(((((int)vz * a.Hi + vy) * a.Wi + vx) * 16 + half * 8) * 2)       // plane-major input
                        : (uint32_t)(((((int)vz * a.Hi + vy) * a.Wi + vx) * a.CPi + (ci_t0 + (int)pl) * 16 + half * 8) * 2);
    crdx[j] = (int)vz | (vy << 8) | (vx << 16);
  }
#pragma unroll
  for (int j = 0; j < WD_NJD; ++j) {
    int c = (wave + 4 * j) * 64 + lane;
    c = c < P.ndz_chunks ? c : P.ndz_chunks - 1;
    const int half = c & 1, rest = c >> 1;
    const uint32_t pl = fdiv(rest, P.d_tv);
    const uint32_t vox = rest - pl * P.TV;
    const int rx = vox & 31, row = vox >> 5;
    const uint32_t rz = fdiv(row, P.d_ty_rows);
    const int ry = row - rz * P.TY;
    reld[j] = (uint32_t)(((((int)rz * a.Ho + ry) * a.Wo + rx) * a.CPo + (co_t0 + (int)pl) * 16 + half * 8) * 2);
    crdd[j] = (int)rz | (ry << 8) | (rx << 16);
  }

  // per-lane read offsets for the two transposed reads of a K step (quad = 2g+rd for even g, 2g+1-rd for odd g)
  const int vq0 = ((lg & 1) ? 2 * lg + 1 : 2 * lg) * 4 + lq;
  const int vq1 = ((lg & 1) ? 2 * lg : 2 * lg + 1) * 4 + lq;
  const int off0 = vq0 * WD_VSB + lp * 8, off1 = vq1 * WD_VSB + lp * 8;
  const int offb0 = vq0 * a.sW * WD_VSB + lp * 8, offb1 = vq1 * a.sW * WD_VSB + lp * 8;      // input voxel of output voxel v: v * stride + tap
  // taps of this wave: exactly WD_TW, indices clamped (the duplicate is dropped at the flush)
  int tapoff[WD_TW];
#pragma unroll
  for (int t = 0; t < WD_TW; ++t) {
    int ti = wave * WD_TW + t;
    ti = ti < a.ntap ? ti : a.ntap - 1;
    const int* tp = a.taps + ti * 3;
    tapoff[t] = ((tp[0] * P.XH + tp[1]) * P.XW + tp[2]) * WD_VSB;
  }

  f32x4 acc[WD_TW][COB][CIB];
#pragma unroll
  for (int t = 0; t < WD_TW; ++t)
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
      for (int i = 0; i < CIB; ++i) acc[t][c][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int xplane = P.XV * WD_VSB, dzplane = P.TV * WD_VSB;
  __syncthreads();

  // issue the DMAs of one tile into buffer `buf`
  auto issue = [&](uint32_t tile, int buf) {
    uint32_t t = tile;
    uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
    q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; t = q;
    q = fdiv(t, P.d_tz); const int tz = t - q * P.ntz; const int b = q;
    const int oz0 = tz * P.TZ, oy0 = ty * P.TY, ox0 = tx * 32;
    // input origin of the tile: output origin + o0 (o0 = -padding <= 0: the origin may lie before the volume; such
    // chunks are zero-filled below and their addresses never dereferenced)
    const int iz0 = oz0 * a.sD + a.o0D, iy0 = oy0 * a.sH + a.o0H, ix0 = ox0 * a.sW + a.o0W;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(xg) +
                              (a.x_plane ? (((((int64_t)b * a.Di + iz0) * a.Hi + iy0) * a.Wi + ix0) * 16 + (int64_t)ci_t0 * a.x_plane) * 2
                                         : ((((int64_t)b * a.Di + iz0) * a.Hi + iy0) * a.Wi + ix0) * a.CPi * 2);
    const unsigned char* db = reinterpret_cast<const unsigned char*>(dzg + ((((size_t)b * a.Do + oz0) * a.Ho + oy0) * a.Wo + ox0) * a.CPo);
    const bool interior = oz0 + P.TZ <= a.Do && oy0 + P.TY <= a.Ho && ox0 + 32 <= a.Wo &&
                          iz0 >= 0 && iy0 >= 0 && ix0 >= 0 && iz0 + P.XD <= a.Di && iy0 + P.XH <= a.Hi && ix0 + P.XW <= a.Wi;
    unsigned char* base = lds + buf * P.buf_bytes;
    unsigned char* dbase = base + P.dz_off;
    if (interior) {
#pragma unroll
      for (int j = 0; j < WD_NJX; ++j)
        if (j < P.njx) sp_dma16(xb + relx[j], base + (wave + 4 * j) * 1024);
#pragma unroll
      for (int j = 0; j < WD_NJD; ++j)
        if (j < P.njd) sp_dma16(db + reld[j], dbase + (wave + 4 * j) * 1024);
    } else {
      // border tile: chunks whose voxel lies outside the output grid (dz) or outside the input volume (x) are zeros
#pragma unroll
      for (int j = 0; j < WD_NJX; ++j) {
        if (j < P.njx) {
          const int cz = crdx[j] & 0xff, cy = (crdx[j] >> 8) & 0xff, cx = (crdx[j] >> 16) & 0xff;
          unsigned char* dst = base + (wave + 4 * j) * 1024;
          if ((unsigned)(iz0 + cz) < (unsigned)a.Di && (unsigned)(iy0 + cy) < (unsigned)a.Hi && (unsigned)(ix0 + cx) < (unsigned)a.Wi)
            sp_dma16(xb + relx[j], dst);
          else
            *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < WD_NJD; ++j) {
        if (j < P.njd) {
          const int cz = crdd[j] & 0xff, cy = (crdd[j] >> 8) & 0xff, cx = (crdd[j] >> 16) & 0xff;
          unsigned char* dst = dbase + (wave + 4 * j) * 1024;
          if (oz0 + cz < a.Do && oy0 + cy < a.Ho && ox0 + cx < a.Wo)
            sp_dma16(db + reld[j], dst);
          else
            *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(0, 0, 0, 0);
        }
      }
    }
  };

  // tile walk.  XCD-aware when the grid is a multiple of 8: workgroup b runs on XCD b % 8 (round-robin dispatch), so
  // XCD j takes the j-th contiguous eighth of the tile list and its 64-odd resident workgroups sit on neighbouring
  // tiles -- their shared halos hit that XCD's L2 instead of being fetched once per XCD.
  uint32_t tile = blockIdx.x, tend = P.ntiles, tstep = gridDim.x;
  if (P.xcd) {
    const uint32_t xcd = blockIdx.x & 7, per = (P.ntiles + 7) >> 3;
    tile = xcd * per + (blockIdx.x >> 3);
    tend = min(P.ntiles, (xcd + 1) * per);
    tstep = gridDim.x >> 3;
  }
  int cur = 0;
  if (tile < tend) issue(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nrows = P.TZ * P.TY;
#ifdef SP_CONV_STAMPS
  unsigned long long t0, t1, t2, t3, s_issue = 0, s_comp = 0, s_wait = 0, ntl = 0;
#endif
  for (; tile < tend; tile += tstep) {
    const uint32_t nxt = tile + tstep;
#ifdef SP_CONV_STAMPS
    WSTAMP(t0);
#endif
    if (nxt < tend) issue(nxt, cur ^ 1);
#ifdef SP_CONV_STAMPS
    WSTAMP(t1);
#endif

    const unsigned char* xt = lds + cur * P.buf_bytes;
    const unsigned char* dzt = xt + P.dz_off;
    for (int row = 0; row < nrows; ++row) {
      const uint32_t rz = fdiv(row, P.d_ty_rows);
      const int ry = row - rz * P.TY;
      const unsigned char* arow = dzt + row * 32 * WD_VSB;
      const unsigned char* brow = xt + (((int)rz * a.sD * P.XH + ry * a.sH) * P.XW) * WD_VSB;
      bf16x8 af[COB];
#pragma unroll
      for (int c = 0; c < COB; ++c) af[c] = wd_tr_read2(arow + c * dzplane + off0, arow + c * dzplane + off1);
#pragma unroll
      for (int tt = 0; tt < WD_TW; ++tt) {
#pragma unroll
        for (int i = 0; i < CIB; ++i) {
          const unsigned char* bp = brow + i * xplane + tapoff[tt];
          const bf16x8 bf = wd_tr_read2(bp + offb0, bp + offb1);
#pragma unroll
          for (int c = 0; c < COB; ++c)
            acc[tt][c][i] = SP_MFMA16(af[c], bf, acc[tt][c][i], 0, 0, 0);
        }
      }
    }
#ifdef SP_CONV_STAMPS
    WSTAMP(t2);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile has landed (it had the whole compute phase)
    __syncthreads();                                    // and everyone is done reading the current one
#ifdef SP_CONV_STAMPS
    WSTAMP(t3);
    s_issue += t1 - t0; s_comp += t2 - t1; s_wait += t3 - t2; ntl += 1;
#endif
    cur ^= 1;
  }
#ifdef SP_CONV_STAMPS
  if (tid == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    sp_stamp_buf[blockIdx.x][0] = s_issue; sp_stamp_buf[blockIdx.x][1] = s_comp; sp_stamp_buf[blockIdx.x][2] = s_wait;
    sp_stamp_buf[blockIdx.x][3] = ntl;
  }
#endif

  // ---- flush: D[row = co = lg*4+j][col = ci = li] -> dw_acc[tap][co][ci] ---------------------------------------
  const int CoP = a.CoT * 16, CiP = a.CiT * 16;
  // parts mode: this workgroup's own block of partial sums (plain stores), else fp32 atomics into the one block
  float* prow = a.dw_acc + (a.parts ? (size_t)blockIdx.x * a.ntap * CoP * CiP : (size_t)0);
#pragma unroll
  for (int tt = 0; tt < WD_TW; ++tt) {
    const int tap = wave * WD_TW + tt;
    if (tap < a.ntap) {
#pragma unroll
      for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int i = 0; i < CIB; ++i) {
          if (co_t0 + c < a.CoT && ci_t0 + i < a.CiT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int co = (co_t0 + c) * 16 + lg * 4 + j, ci = (ci_t0 + i) * 16 + li;
              float* dst = prow + ((size_t)tap * CoP + co) * CiP + ci;
              if (a.parts) *dst = acc[tt][c][i][j]; else atomicAdd(dst, acc[tt][c][i][j]);
            }
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// z-marching variant (one 16-channel input plane per workgroup, COB <= 2): a workgroup owns a column of TY = 8 output
// rows x 32 voxels and walks it in z with the input in a RING of four z-plane slots ((TY + 2) x 34 voxels) and the
// output gradient in two plane buffers.  Per output plane ONE new x plane and one dz plane are fetched (the tiled
// kernel above re-reads 3.2x the input per tile and issues 1.8x the DMAs per MFMA); the fetch for plane z + 1 is
// issued before the MFMAs of plane z.
struct WgradZsDev {
  sp_wgrad_args a;
  int32_t TY, XH, XW, XPB, nxc, ndc, nty, ntx, nzc, ZC, xcd;
  uint32_t ncols;
  FastDiv d_tx, d_ty, d_zc, d_xw;
};

template <int COB>
__global__ __launch_bounds__(256, 2) void wgrad_zs_kernel(const WgradZsDev P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int NJX = 3, NJD = 2 * COB, DZB = 8 * 32 * 32;     // chunks per lane: x plane, dz plane(s); bytes of one dz plane
  const sp_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int co_t0 = blockIdx.y * COB, ci_t0 = blockIdx.z;
  unsigned char* xring = lds;
  unsigned char* dzb = lds + 4 * P.XPB;
  const int xpitch = a.x_plane ? 16 : a.CPi;

  // ---- per-lane DMA plans (column-invariant)
  uint32_t relx[NJX], reld[NJD];
  int crdx[NJX], crdd[NJD];
#pragma unroll
  for (int j = 0; j < NJX; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;
    const int cc = c < P.nxc ? c : P.nxc - 1;
    const int half = cc & 1, vox = cc >> 1;
    const int vy = fdiv(vox, P.d_xw), vx = vox - vy * P.XW;
    relx[j] = (uint32_t)(((vy * a.Wi + vx) * xpitch + (a.x_plane ? 0 : ci_t0 * 16) + half * 8) * 2);
    crdx[j] = vy | (vx << 8) | (c < P.nxc ? 0 : (1 << 30));
  }
#pragma unroll
  for (int j = 0; j < NJD; ++j) {
    const int c = (wave + 4 * j) * 64 + lane;           // < COB * 512 always
    const int half = c & 1, rest = c >> 1, pl = rest >> 8, vox = rest & 255, ry = vox >> 5, rx = vox & 31;
    reld[j] = (uint32_t)(((ry * a.Wo + rx) * a.CPo + (co_t0 + pl) * 16 + half * 8) * 2);
    crdd[j] = ry | (rx << 8) | ((co_t0 + pl < a.CoT) ? 0 : (1 << 30));
  }
  const int vq0 = ((lg & 1) ? 2 * lg + 1 : 2 * lg) * 4 + lq;
  const int vq1 = ((lg & 1) ? 2 * lg : 2 * lg + 1) * 4 + lq;
  const int off0 = vq0 * WD_VSB + lp * 8, off1 = vq1 * WD_VSB + lp * 8;
  int tapz[WD_TW], tapin[WD_TW];
#pragma unroll
  for (int t = 0; t < WD_TW; ++t) {
    int ti = wave * WD_TW + t;
    ti = ti < a.ntap ? ti : a.ntap - 1;
    const int* tp = a.taps + ti * 3;
    tapz[t] = __builtin_amdgcn_readfirstlane(tp[0]);
    tapin[t] = (tp[1] * P.XW + tp[2]) * WD_VSB;
  }
  f32x4 acc[WD_TW][COB];
#pragma unroll
  for (int t = 0; t < WD_TW; ++t)
#pragma unroll
    for (int c = 0; c < COB; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Work = (column, output plane) pairs, cols_xy * Do of them, cut into gridDim.x equal pieces: a workgroup takes the
  // planes [vb*T/N, (vb+1)*T/N) of the flattened sequence -- perfect balance for any volume (a piece that crosses a
  // column boundary pays one more three-plane prologue).  vb: XCD-aware id, neighbours in the sequence share an L2.
  const uint32_t vb = P.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
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
    const int oy0 = ty * P.TY, ox0 = tx * 32;
    const int iy0 = oy0 + a.o0H, ix0 = ox0 + a.o0W;
    int xmask = 0, dmask = 0;
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int vy = crdx[j] & 0xff, vx = (crdx[j] >> 8) & 0xff;
      if (!(crdx[j] >> 30) && (unsigned)(iy0 + vy) < (unsigned)a.Hi && (unsigned)(ix0 + vx) < (unsigned)a.Wi) xmask |= 1 << j;
    }
#pragma unroll
    for (int j = 0; j < NJD; ++j) {
      const int ry = crdd[j] & 0xff, rx = (crdd[j] >> 8) & 0xff;
      if (!(crdd[j] >> 30) && oy0 + ry < a.Ho && ox0 + rx < a.Wo) dmask |= 1 << j;
    }
    const unsigned char* xcol = reinterpret_cast<const unsigned char*>(a.x) +
        ((((int64_t)b * a.Di) * a.Hi + iy0) * a.Wi + ix0) * xpitch * 2 + (a.x_plane ? (int64_t)ci_t0 * a.x_plane * 2 : 0);
    const unsigned char* dcol = reinterpret_cast<const unsigned char*>(a.dz) + ((((int64_t)b * a.Do) * a.Ho + oy0) * a.Wo + ox0) * a.CPo * 2;
    auto load_x = [&](int iz, int slot) {
      const bool zin = (unsigned)iz < (unsigned)a.Di;
      const unsigned char* src0 = xcol + (int64_t)iz * a.Hi * a.Wi * xpitch * 2;
      unsigned char* dst0 = xring + slot * P.XPB;
#pragma unroll
      for (int j = 0; j < NJX; ++j) {
        unsigned char* dst = dst0 + (wave + 4 * j) * 1024;
        if (!(crdx[j] >> 30)) {
          if (zin && ((xmask >> j) & 1)) sp_dma16(src0 + relx[j], dst);
          else *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(0, 0, 0, 0);
        }
      }
    };
    auto load_dz = [&](int oz, int buf) {
      const unsigned char* src0 = dcol + (int64_t)oz * a.Ho * a.Wo * a.CPo * 2;
      unsigned char* dst0 = dzb + buf * COB * DZB;
#pragma unroll
      for (int j = 0; j < NJD; ++j) {
        unsigned char* dst = dst0 + (wave + 4 * j) * 1024;
        if ((dmask >> j) & 1) sp_dma16(src0 + reld[j], dst);
        else *reinterpret_cast<uint4*>(dst + lane * 16) = make_uint4(0, 0, 0, 0);
      }
    };
    __syncthreads();                                      // the previous column has been consumed
    load_x(z0 + a.o0D, 0); load_x(z0 + a.o0D + 1, 1); load_x(z0 + a.o0D + 2, 2);
    load_dz(z0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int z = z0; z < z1; ++z) {
      const int sl = (z - z0) & 3, db = (z - z0) & 1;
      if (z + 1 < z1) { load_x(z + a.o0D + 3, (sl + 3) & 3); load_dz(z + 1, db ^ 1); }
      const int sb0 = sl * P.XPB, sb1 = ((sl + 1) & 3) * P.XPB, sb2 = ((sl + 2) & 3) * P.XPB;
      int tb[WD_TW];
#pragma unroll
      for (int tt = 0; tt < WD_TW; ++tt) tb[tt] = (tapz[tt] == 0 ? sb0 : (tapz[tt] == 1 ? sb1 : sb2)) + tapin[tt];
      const unsigned char* dzt = dzb + db * COB * DZB;
      for (int row = 0; row < P.TY; ++row) {
        const unsigned char* arow = dzt + row * 32 * WD_VSB;
        const unsigned char* brow = xring + row * P.XW * WD_VSB;
        bf16x8 af[COB];
#pragma unroll
        for (int c = 0; c < COB; ++c) af[c] = wd_tr_read2(arow + c * DZB + off0, arow + c * DZB + off1);
#pragma unroll
        for (int tt = 0; tt < WD_TW; ++tt) {
          const unsigned char* bp = brow + tb[tt];
          const bf16x8 bf = wd_tr_read2(bp + off0, bp + off1);
#pragma unroll
          for (int c = 0; c < COB; ++c) acc[tt][c] = SP_MFMA16(af[c], bf, acc[tt][c], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  // ---- flush (parts mode only): this workgroup's block of partial sums
  const int CoP = a.CoT * 16, CiP = a.CiT * 16;
  float* prow = a.dw_acc + (size_t)blockIdx.x * a.ntap * CoP * CiP;
#pragma unroll
  for (int tt = 0; tt < WD_TW; ++tt) {
    const int tap = wave * WD_TW + tt;
    if (tap < a.ntap) {
#pragma unroll
      for (int c = 0; c < COB; ++c)
        if (co_t0 + c < a.CoT) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = (co_t0 + c) * 16 + lg * 4 + j, ci = ci_t0 * 16 + li;
            prow[((size_t)tap * CoP + co) * CiP + ci] = acc[tt][c][j];
          }
        }
    }
  }
}

static int launch_wgrad_zs(const sp_wgrad_args* a, int COB, hipStream_t st) {
  WgradZsDev P;
  P.a = *a;
  P.TY = 8; P.XH = P.TY + a->kH - 1; P.XW = 31 + a->kW;
  P.nxc = P.XH * P.XW * 2;
  P.XPB = (P.nxc * 16 + 1023) / 1024 * 1024;
  if (P.nxc > 3 * 256 || a->kD != 3 || a->kH > 3 || a->kW > 3) return 1;      // not this variant
  P.ndc = COB * 512;
  P.nty = (a->Ho + P.TY - 1) / P.TY; P.ntx = (a->Wo + 31) / 32;
  const int cols_xy = a->B * P.nty * P.ntx;
  P.nzc = 1; P.ZC = a->Do;
  P.ncols = (uint32_t)cols_xy;
  P.d_tx = make_fastdiv(P.ntx); P.d_ty = make_fastdiv(P.nty); P.d_zc = make_fastdiv(P.nzc); P.d_xw = make_fastdiv(P.XW);
  const uint32_t gx = a->nblocks;
  P.xcd = (gx % 8 == 0 && gx >= 8 && !getenv("SP_WGRAD_NOXCD")) ? 1 : 0;
  const int lds_bytes = 4 * P.XPB + 2 * COB * 8192;
  dim3 grid(gx, (a->CoT + COB - 1) / COB, a->CiT);
#define WZ_CASE(C_)                                                                                  \
  if (COB == C_) {                                                                                   \
    auto kern = wgrad_zs_kernel<C_>;                                                                 \
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv");                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, P);                                     \
    SP_CHECK_LAUNCH("sp_conv3d_wgrad(zs)");                                                          \
    return SP_OK;                                                                                    \
  }
  WZ_CASE(1) WZ_CASE(2)
#undef WZ_CASE
  return 1;
}

int sp_wgrad_zr_try(const sp_wgrad_args* a, hipStream_t st);   // sp_wgrad_zr.hip

int sp_conv3d_wgrad_dma(const sp_wgrad_args* a, sp_stream_t stream) {
  SP_CHECK_ARG(a->dtype == SP_BF16 && !a->in_scale && !a->dz_scale, "sp_conv3d_wgrad(dma): bf16, no affine on load");
  const bool unit = a->sD == 1 && a->sH == 1 && a->sW == 1;
  SP_CHECK_ARG(a->sD >= 1 && a->sD <= 2 && a->sH >= 1 && a->sH <= 2 && a->sW >= 1 && a->sW <= 2 &&
               a->o0D <= 0 && a->o0H <= 0 && a->o0W <= 0 && a->o0D >= -2 && a->o0H >= -2 && a->o0W >= -2,
               "sp_conv3d_wgrad(dma): stride 1 or 2, padding 0..2 only");
  SP_CHECK_ARG(a->ntap >= 1 && a->ntap <= 28 && a->kD >= 1 && a->kD <= 3 && a->kH >= 1 && a->kH <= 3 && a->kW >= 1 && a->kW <= 3, "sp_conv3d_wgrad(dma): at most 3x3x3 = 28 taps (7 per wave)");
  // every tap of every output voxel lies inside the zero-padded input (a strided convolution may leave a remainder unused)
  SP_CHECK_ARG(a->Di - 2 * a->o0D >= (a->Do - 1) * a->sD + a->kD && a->Hi - 2 * a->o0H >= (a->Ho - 1) * a->sH + a->kH && a->Wi - 2 * a->o0W >= (a->Wo - 1) * a->sW + a->kW,
               "sp_conv3d_wgrad(dma): input / output extents do not match a convolution with this stride and padding");
  SP_CHECK_ARG(!unit || (a->Di - 2 * a->o0D == a->Do + a->kD - 1 && a->Hi - 2 * a->o0H == a->Ho + a->kH - 1 && a->Wi - 2 * a->o0W == a->Wo + a->kW - 1),
               "sp_conv3d_wgrad(dma): input / output extents do not match a stride-1 convolution with this padding");
  // row-sliding z-marching variant (sp_wgrad_zr.hip): un-padded layers with per-workgroup partial blocks
  {
    const int rc = sp_wgrad_zr_try(a, reinterpret_cast<hipStream_t>(stream));
    if (rc <= 0) return rc;
  }
  WgradDmaDev P;
  P.a = *a;
  int COB = a->CoT >= 4 ? 4 : (a->CoT >= 2 ? 2 : 1);
  int CIB = a->CiT >= 3 ? 3 : a->CiT;
  if (COB == 4) CIB = 1;
  if (a->cib > 0 && a->cib <= CIB) CIB = a->cib;          // caller's blocking of the cin tiles (grid.z grows accordingly)
  // z-marching variant: one input plane per workgroup, partial-row flush, enough rows for its 8-row columns
  // (measured: two output planes per workgroup gain 30-40 %, one plane 5 % on the largest volume and nothing below)
  if (unit && a->ntap > 21 && CIB == 1 && a->parts && a->Ho >= 8 && a->Do >= 4 && a->zs &&
      (COB == 2 || (COB == 1 && a->zs >= 2) || (COB == 1 && (int64_t)a->B * a->Do * a->Ho * a->Wo >= 5000000))) {
    const int rc = launch_wgrad_zs(a, COB, reinterpret_cast<hipStream_t>(stream));
    if (rc <= 0) return rc;
  }
  // tile rows: the largest TZ x TY whose two buffers fit 150 KiB and whose chunk count fits the per-lane plan
  SP_CHECK_ARG(a->CPi % 16 == 0 && a->CPo % 16 == 0, "sp_conv3d_wgrad(dma): channel pitches must be multiples of 16");
  SP_CHECK_ARG(a->x_plane == 0 || a->cib == 1, "sp_conv3d_wgrad(dma): plane-major input needs one input plane per workgroup (cib = 1)");
  // tile rows: two buffers per workgroup; prefer a shape that lets TWO workgroups share a CU (<= 75 KiB each) so
  // that every SIMD has two waves to overlap LDS latency with MFMA issue
  static const int cand[][2] = {{4, 4}, {4, 2}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
  bool found = false;
  for (int pass = 0; pass < 2 && !found; ++pass) {
    for (auto& cz : cand) {
      if (a->tile_rows > 0 && cz[0] * cz[1] != a->tile_rows) continue;
      P.TZ = cz[0]; P.TY = cz[1];
      P.XD = (P.TZ - 1) * a->sD + a->kD; P.XH = (P.TY - 1) * a->sH + a->kH; P.XW = 31 * a->sW + a->kW;
      if (P.XD > 255 || P.XH > 255 || P.XW > 255) continue;      // (packed tile coordinates of the border path)
      P.XV = P.XD * P.XH * P.XW; P.TV = P.TZ * P.TY * 32;
      P.nx_chunks = CIB * P.XV * 2; P.ndz_chunks = COB * P.TV * 2;
      P.njx = (P.nx_chunks + 255) / 256; P.njd = (P.ndz_chunks + 255) / 256;
      P.dz_off = P.njx * 256 * 16;
      P.buf_bytes = P.dz_off + P.njd * 256 * 16;
      const int limit = (pass == 0 && a->tile_rows == 0) ? 75 * 1024 : 150 * 1024;
      if (P.njx <= WD_NJX && P.njd <= WD_NJD && 2 * P.buf_bytes <= limit) { found = true; break; }
    }
  }
  SP_CHECK_ARG(found, "sp_conv3d_wgrad(dma): no tile fits");
  P.ntx = (a->Wo + 31) / 32; P.nty = (a->Ho + P.TY - 1) / P.TY; P.ntz = (a->Do + P.TZ - 1) / P.TZ;
  P.d_tx = make_fastdiv(P.ntx); P.d_ty = make_fastdiv(P.nty); P.d_tz = make_fastdiv(P.ntz);
  P.d_xw = make_fastdiv(P.XW); P.d_xh = make_fastdiv(P.XH); P.d_xv = make_fastdiv(P.XV); P.d_tv = make_fastdiv(P.TV);
  P.d_ty_rows = make_fastdiv(P.TY);
  const uint64_t nt = (uint64_t)P.ntx * P.nty * P.ntz * a->B;
  SP_CHECK_ARG(nt < (1ull << 31), "sp_conv3d_wgrad(dma): too many tiles");
  P.ntiles = (uint32_t)nt;
  const int lds_bytes = 2 * P.buf_bytes;
  uint32_t gx = (a->parts || a->nblocks < (int64_t)nt) ? a->nblocks : (uint32_t)nt;   // parts: every block is written
  dim3 grid(gx, (a->CoT + COB - 1) / COB, (a->CiT + CIB - 1) / CIB);
  P.xcd = (gx % 8 == 0 && gx >= 8 && !getenv("SP_WGRAD_NOXCD")) ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define WD_CASE(C_, I_)                                                                              \
  if (COB == C_ && CIB == I_) {                                                                      \
    auto kern = wgrad_dma_kernel<C_, I_>;                                                            \
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv");                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, P);                                     \
    SP_CHECK_LAUNCH("sp_conv3d_wgrad(dma)");                                                         \
    return SP_OK;                                                                                    \
  }
  WD_CASE(1, 1) WD_CASE(1, 2) WD_CASE(1, 3) WD_CASE(2, 1) WD_CASE(2, 2) WD_CASE(2, 3) WD_CASE(4, 1)
#undef WD_CASE
  sp_set_error("sp_conv3d_wgrad(dma): no kernel for COB=%d CIB=%d", COB, CIB);
  return SP_EINVAL;
}

// BatchNorm-backward sums of a layer whose input gradient is not needed (the network's first layer), taken from the
// weight gradient instead of running the data-gradient convolution:  with g = conv_transpose(dz, W) over an
// un-padded convolution,  sum_v g[v,ci] = sum_{co,tap} W[co,ci,tap] * sum_v dz[v,co]   and
// sum_v g[v,ci]*x[v,ci] = sum_{co,tap} W[co,ci,tap] * acc[tap][co][ci]   (acc = the raw-input weight gradient).
__device__ __forceinline__ void bn_sums_from_wgrad(float w, float acc, double dbias_co, double* sums2) {
  atomicAdd(&sums2[0], (double)w * dbias_co);
  atomicAdd(&sums2[1], (double)w * (double)acc);
}

// parts mode of the folded finish (same 32-entry x 8-row-lane reduction as wgrad_finish_parts_kernel)
template <int RL>      // row lanes: threads that share an entry and split the partial blocks (8 -> 32: 10 -> ~5 us for 512 blocks)
__global__ __launch_bounds__(32 * RL) void wgrad_finish_folded_parts_kernel(const float* __restrict__ acc, int nparts,
                                                                        const int32_t* __restrict__ tapsrc, int ntap,
                                                                        int CoP, int CiP, int Cout, int Cin, int64_t sCo,
                                                                        int64_t sCi, const float* __restrict__ scale,
                                                                        const float* __restrict__ shift,
                                                                        const double* __restrict__ dbias,
                                                                        float* __restrict__ dw, float* __restrict__ dbias_grad,
                                                                        const float* __restrict__ wbn,
                                                                        double* __restrict__ bn_sums, int bn_nrep, int bn_cp, int dbs,
                                                                        float acc_scale) {
  __shared__ float red[RL][33];
  const int64_t total = (int64_t)ntap * CoP * CiP;
  const int64_t gid = (int64_t)blockIdx.x * (32 * RL) + threadIdx.x;
  if (dbias_grad && gid < Cout) dbias_grad[gid] += (float)sp_rows_sum(dbias, (int)gid, dbs);
  const int el = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t idx = (int64_t)blockIdx.x * 32 + el;
  float s = 0.f;
  if (idx < total) {
    int r = rl;
    for (; r + 3 * RL < nparts; r += 4 * RL) {
      const float a0 = acc[(size_t)r * total + idx], a1 = acc[(size_t)(r + RL) * total + idx];
      const float a2 = acc[(size_t)(r + 2 * RL) * total + idx], a3 = acc[(size_t)(r + 3 * RL) * total + idx];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; r < nparts; r += RL) s += acc[(size_t)r * total + idx];
  }
  red[rl][el] = s;
  __syncthreads();
  if (rl != 0 || idx >= total) return;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < RL; ++i) v += red[i][el];
  v *= acc_scale;                  // (fp8 weight gradient: the accumulators carry the power-of-two scale of the quantised dz)
  const int ci = idx % CiP;
  const int co = (idx / CiP) % CoP;
  const int t = idx / ((int64_t)CiP * CoP);
  if (co < Cout && ci < Cin) {
    const int64_t wi = co * sCo + ci * sCi + tapsrc[t];
    const double db = sp_rows_sum(dbias, co, dbs);
    dw[wi] += scale[ci] * v + shift[ci] * (float)db;
    if (bn_sums) bn_sums_from_wgrad(wbn[wi], v, db, bn_sums + (size_t)(blockIdx.x % bn_nrep) * bn_cp * 2 + ci * 2);
  }
}

// The same for FEW, LARGE partial blocks (the 64..384-channel layers of the 4-scale network: 8-32 blocks of 1.7-7 MB).  The kernel
// above takes 32 consecutive accumulator entries per workgroup = 32 input channels of ONE tap, and its read-modify-write of dw
// then touches 32 different lines (dw is [co][ci][tap]: 27 floats apart) -- 1.8 M scattered 4-byte updates for 256 -> 256, 284 us
// per launch, 2 ms of the fp8 step (profiles/r03_unet4_fp8_kernel_stats.csv).  Here a workgroup owns (one output channel, 32 input
// channels, ALL taps): the accumulator rows come in as 128-byte pieces per (tap, block), the tile is transposed through LDS and
// leaves as ONE contiguous 32 x ntap run of dw; the BatchNorm-backward sums are reduced over the taps first (2 atomics per input
// channel and workgroup instead of 2 per entry).
template <int NTAP>
__global__ __launch_bounds__(256) void wgrad_finish_folded_tile_kernel(const float* __restrict__ acc, int nparts,
                                                                       const int32_t* __restrict__ tapsrc, int CoP, int CiP, int Cout,
                                                                       int Cin, int64_t sCo, int64_t sCi, const float* __restrict__ scale,
                                                                       const float* __restrict__ shift, const double* __restrict__ dbias,
                                                                       float* __restrict__ dw, float* __restrict__ dbias_grad,
                                                                       const float* __restrict__ wbn, double* __restrict__ bn_sums,
                                                                       int bn_nrep, int bn_cp, int dbs, float acc_scale) {
  __shared__ float tile[NTAP][33];
  const int el = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int ci0 = blockIdx.x * 32, co = blockIdx.y;
  const int64_t total = (int64_t)NTAP * CoP * CiP;
  if (dbias_grad && blockIdx.x == 0 && threadIdx.x == 0 && co < Cout) dbias_grad[co] += (float)sp_rows_sum(dbias, co, dbs);
  const bool civ = ci0 + el < CiP;
  for (int t = rl; t < NTAP; t += 8) {
    const float* p = acc + ((int64_t)t * CoP + co) * CiP + ci0 + el;
    float s0 = 0.f, s1 = 0.f;
    if (civ) {
      int r = 0;
      for (; r + 1 < nparts; r += 2) { s0 += p[(size_t)r * total]; s1 += p[(size_t)(r + 1) * total]; }
      if (r < nparts) s0 += p[(size_t)r * total];
    }
    tile[t][el] = (s0 + s1) * acc_scale;
  }
  __syncthreads();
  if (co >= Cout) return;
  const double db = sp_rows_sum(dbias, co, dbs);
  const float dbf = (float)db;
  // 32 x NTAP values of dw[co][ci0 .. ci0+31][:] in memory order (contiguous when sCi == NTAP and tapsrc is the identity)
  for (int k = threadIdx.x; k < 32 * NTAP; k += 256) {
    const int i = k / NTAP, t = k - i * NTAP;
    const int ci = ci0 + i;
    if (ci < Cin) {
      const int64_t wi = co * sCo + ci * sCi + tapsrc[t];
      dw[wi] += scale[ci] * tile[t][i] + shift[ci] * dbf;
    }
  }
  if (bn_sums && threadIdx.x < 32 && ci0 + (int)threadIdx.x < Cin) {      // (sum g, sum g*x) of input channel ci: reduced over the taps first
    const int ci = ci0 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 9
    for (int t = 0; t < NTAP; ++t) {
      const float w = wbn[co * sCo + ci * sCi + tapsrc[t]];
      a0 += w;
      a1 = fmaf(w, tile[t][threadIdx.x], a1);
    }
    double* dst = bn_sums + (size_t)((blockIdx.x + blockIdx.y) % bn_nrep) * bn_cp * 2 + ci * 2;
    atomicAdd(dst, (double)a0 * db);
    atomicAdd(dst + 1, (double)a1);
  }
}

// dw[co,ci,tap] += scale[ci]*acc[tap][co][ci] + shift[ci]*dbias[co]   (BatchNorm folded out of the operand load)
__global__ void wgrad_finish_folded_kernel(float* __restrict__ acc, const int32_t* __restrict__ tapsrc, int ntap,
                                           int CoP, int CiP, int Cout, int Cin, int64_t sCo, int64_t sCi,
                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                           const double* __restrict__ dbias, float* __restrict__ dw,
                                           float* __restrict__ dbias_grad, const float* __restrict__ wbn,
                                           double* __restrict__ bn_sums, int bn_nrep, int bn_cp, int dbs, float acc_scale) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)ntap * CoP * CiP;
  if (dbias_grad && idx < Cout) dbias_grad[idx] += (float)sp_rows_sum(dbias, (int)idx, dbs);
  if (idx >= total) return;
  const int ci = idx % CiP;
  const int co = (idx / CiP) % CoP;
  const int t = idx / ((int64_t)CiP * CoP);
  const float v = acc[idx] * acc_scale;
  acc[idx] = 0.f;
  if (co < Cout && ci < Cin) {
    const int64_t wi = co * sCo + ci * sCi + tapsrc[t];
    const double db = sp_rows_sum(dbias, co, dbs);
    dw[wi] += scale[ci] * v + shift[ci] * (float)db;
    if (bn_sums) bn_sums_from_wgrad(wbn[wi], v, db, bn_sums + (size_t)(blockIdx.x % bn_nrep) * bn_cp * 2 + ci * 2);
  }
}

static int wgrad_finish_folded_impl(float* dw_acc, int32_t nparts, const int32_t* tapsrc, int32_t ntap, int32_t CoP, int32_t CiP,
                                    int32_t Cout, int32_t Cin, int64_t sCo, int64_t sCi, const float* scale,
                                    const float* shift, const double* dbias_sums, float* dw, float* dbias_grad,
                                    const float* w_for_bn, double* bn_sums, int32_t bn_nrep, int32_t bn_cp, int32_t dbias_stride,
                                    float acc_scale, sp_stream_t stream) {
  SP_CHECK_ARG(dw_acc && tapsrc && dw && scale && shift && dbias_sums && Cout <= CoP && Cin <= CiP, "sp_wgrad_finish_folded: bad arguments");
  const int64_t total = (int64_t)ntap * CoP * CiP;
  SP_CHECK_ARG(nparts >= 1 && Cout <= (total + 31) / 32 * 256, "sp_wgrad_finish_folded: nparts");      // (dbias_grad: one thread per output channel)
  SP_CHECK_ARG(!bn_sums || (w_for_bn && bn_nrep >= 1), "sp_wgrad_finish_folded: bn_sums needs the weights and a replica count");
  if (bn_cp <= 0) bn_cp = CiP;
  static const int tile_max_ = getenv("SP_WGRAD_FINISH_TILE") ? atoi(getenv("SP_WGRAD_FINISH_TILE")) : 32;      // (0: off; A/B knob)
  if (nparts > 1 && nparts <= tile_max_ && ntap == 27 && CiP % 32 == 0 && (int64_t)CoP * (CiP / 32) >= 512) {
    hipLaunchKernelGGL(wgrad_finish_folded_tile_kernel<27>, dim3((unsigned)(CiP / 32), (unsigned)CoP), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), dw_acc, nparts, tapsrc, CoP, CiP, Cout, Cin, sCo, sCi, scale, shift,
                       dbias_sums, dw, dbias_grad, w_for_bn, bn_sums, bn_nrep, bn_cp, dbias_stride, acc_scale);
    SP_CHECK_LAUNCH("sp_wgrad_finish_folded");
    return SP_OK;
  }
  if (nparts > 1) {
    if (nparts >= 128)
      hipLaunchKernelGGL(wgrad_finish_folded_parts_kernel<32>, dim3((unsigned)((total + 31) / 32)), dim3(1024), 0,
                         reinterpret_cast<hipStream_t>(stream), dw_acc, nparts, tapsrc, ntap, CoP, CiP, Cout, Cin, sCo, sCi,
                         scale, shift, dbias_sums, dw, dbias_grad, w_for_bn, bn_sums, bn_nrep, bn_cp, dbias_stride, acc_scale);
    else
      hipLaunchKernelGGL(wgrad_finish_folded_parts_kernel<8>, dim3((unsigned)((total + 31) / 32)), dim3(256), 0,
                         reinterpret_cast<hipStream_t>(stream), dw_acc, nparts, tapsrc, ntap, CoP, CiP, Cout, Cin, sCo, sCi,
                         scale, shift, dbias_sums, dw, dbias_grad, w_for_bn, bn_sums, bn_nrep, bn_cp, dbias_stride, acc_scale);
    SP_CHECK_LAUNCH("sp_wgrad_finish_folded");
    return SP_OK;
  }
  hipLaunchKernelGGL(wgrad_finish_folded_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dw_acc, tapsrc, ntap, CoP, CiP, Cout, Cin, sCo, sCi, scale,
                     shift, dbias_sums, dw, dbias_grad, w_for_bn, bn_sums, bn_nrep, bn_cp, dbias_stride, acc_scale);
  SP_CHECK_LAUNCH("sp_wgrad_finish_folded");
  return SP_OK;
}
// ---- folded finish of a PADDED stride-1 3x3x3 layer whose BatchNorm differs per group of the batch (the CAE's batched passes;
// sp_conv_args.bias_tab is the forward's side of the same fold).  The weight gradient ran on the RAW input x (zero padded), into
// per-workgroup partial blocks whose index order follows the samples (sp_wgrad_zr.hip), so blocks [g nb / G, (g + 1) nb / G) belong
// to group g.  With x^ = s_g x + t_g inside the volume and 0 in the padding:
//   dW[co][ci][tap] += sum_g  s_g[ci] A_g[tap][co][ci] + t_g[ci] Sv_g[tap][co],     A_g = the group's partial blocks added up,
//   Sv_g[tap][co] = sum of dz over the output voxels whose tap lies inside the input = a sum of border classes of cls_sums
//   (sp_bn_act_bwd_groups_cls); dbias[co] += sum_g sum_classes; and the BatchNorm-backward pair of the layer's input, per group,
//   (sum_v g, sum_v g x)[ci] = sum_{tap, co} W[co][ci][tap] (Sv_g[tap][co], A_g[tap][co][ci])   (g = the data gradient: it needs no
//   statistics epilogue and no second read of x then).
__device__ __forceinline__ bool fg_tap_valid(int t, int c, int p) { return c < p ? t >= p - c : (c == p ? true : t < 3 - (c - p)); }
// a workgroup = 32 consecutive accumulator entries x 8 row lanes that split the group's partial blocks (a serial walk over the
// blocks per entry, 27 workgroups in all, took 116 us of dependent loads)
__global__ __launch_bounds__(256) void wgrad_finish_folded_groups_kernel(const float* __restrict__ acc, int nb, int ntap, int G, int CoP, int CiP, int Cout, int Cin,
                                                                          int64_t sCo, int64_t sCi, const float* __restrict__ coef, int coef_gstride,
                                                                          int coef_pitch, const double* __restrict__ cls, int pz, int py, int px,
                                                                          const float* __restrict__ w, float* __restrict__ dw, float* __restrict__ dbias_grad,
                                                                          double* __restrict__ bn_sums, int bn_nrep, int bn_cp) {
  __shared__ float red[8][33];
  __shared__ double bred[2][32];
  const int el = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t total = (int64_t)ntap * CoP * CiP;
  const int64_t e = (int64_t)blockIdx.x * 32 + el;
  const bool in = e < total;
  const int ci = in ? (int)(e % CiP) : 0, co = in ? (int)((e / CiP) % CoP) : 0, tap = in ? (int)(e / ((int64_t)CiP * CoP)) : 0;
  const int ny = 2 * py + 1, nx = 2 * px + 1, ncls = (2 * pz + 1) * ny * nx;      // (ntap == 1: padding 0, one class)
  const int per = nb / G;
  const bool real = in && co < Cout && ci < Cin;
  const float wv = (real && rl == 0) ? w[co * sCo + ci * sCi + tap] : 0.f;
  // Sv[g][pair]: the class sums whose tap lies inside the input, for the (tap, co) pairs of this workgroup's 32 entries (at most
  // 32 / CiP + 1 of them): a row lane takes a (group, pair), its 32 lanes the classes (one thread walking 75 classes x G groups of
  // fp64 loads cost 60 us)
  __shared__ double svs[16][8], sva[16][8];
  const int64_t e0 = (int64_t)blockIdx.x * 32;
  const int pair0 = (int)(e0 / CiP);                                   // first (tap * CoP + co) of the workgroup
  const int npair = (int)((min(e0 + 31, total - 1)) / CiP) - pair0 + 1;
  for (int q = rl; q < G * npair; q += 8) {
    const int g = q / npair, pr = pair0 + q % npair;
    const int pco = pr % CoP, ptap = pr / CoP;
    const int ptz = ntap == 27 ? ptap / 9 : 1, pty = ntap == 27 ? (ptap / 3) % 3 : 1, ptx = ntap == 27 ? ptap % 3 : 1;      // (pointwise: the centre, always inside)
    double t = 0.0, ta = 0.0;
    for (int cl = el; cl < ncls; cl += 32) {
      const int cx = cl % nx, cy = (cl / nx) % ny, cz = cl / (nx * ny);
      const double v = cls[((size_t)g * ncls + cl) * CoP + pco];
      ta += v;                                                          // (all classes: the bias gradient's sum dz)
      if (fg_tap_valid(ptz, cz, pz) && fg_tap_valid(pty, cy, py) && fg_tap_valid(ptx, cx, px)) t += v;
    }
    for (int o = 16; o > 0; o >>= 1) { t += __shfl_xor(t, o, 32); ta += __shfl_xor(ta, o, 32); }
    if (el == 0 && g < 16 && q % npair < 8) { svs[g][q % npair] = t; sva[g][q % npair] = ta; }
  }
  __syncthreads();
  float dwv = 0.f;
  for (int g = 0; g < G; ++g) {
    float a0 = 0.f, a1 = 0.f;
    if (in) {
      // eight independent loads in flight per thread (a two-load loop waits a memory round trip per iteration: 16 us per group)
      const float* p = acc + (size_t)g * per * total + e;
      for (int b = rl; b < per; b += 64) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (b + 8 * k < per) ? p[(size_t)(b + 8 * k) * total] : 0.f;
        a0 += (t[0] + t[2]) + (t[4] + t[6]);
        a1 += (t[1] + t[3]) + (t[5] + t[7]);
      }
    }
    red[rl][el] = a0 + a1;
    __syncthreads();
    if (rl == 0) {
      float A = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) A += red[r][el];
      const double Sv = in ? svs[g][(int)(e / CiP) - pair0] : 0.0;
      const float* cf = coef + (size_t)g * coef_gstride;
      if (real) dwv += cf[ci] * A + cf[2 * coef_pitch + ci] * (float)Sv;
      bred[0][el] = (double)wv * Sv;
      bred[1][el] = (double)wv * (double)A;
      if (dbias_grad && in && tap == 0 && ci == 0 && co < Cout) atomicAdd(&dbias_grad[co], (float)sva[g][(int)(e / CiP) - pair0]);
    }
    __syncthreads();
    // the BatchNorm-backward pair: this workgroup's entries of one input channel first (32 consecutive entries: ci repeats with
    // period CiP), then one atomic per (workgroup, ci) into a replica row
    if (bn_sums && rl == 0) {
      const int c0 = (int)(((int64_t)blockIdx.x * 32) % CiP);
      // a lane plays the channels el, el + 32, ... (CiP > 32: the 48- and 64-channel tiles; one pass otherwise)
      for (int ch = el; ch < CiP && ch < Cin; ch += 32) {
        const int first = ((ch - c0) % CiP + CiP) % CiP;
        if (first >= 32) continue;
        double s0 = 0.0, s1 = 0.0;
        for (int k = first; k < 32; k += CiP) { s0 += bred[0][k]; s1 += bred[1][k]; }
        double* o = bn_sums + ((size_t)g * bn_nrep + (blockIdx.x % bn_nrep)) * bn_cp * 2 + (size_t)ch * 2;
        atomicAdd(&o[0], s0);
        atomicAdd(&o[1], s1);
      }
    }
    __syncthreads();
  }
  if (real && rl == 0) dw[co * sCo + ci * sCi + tap] += dwv;
}
extern "C" int sp_wgrad_finish_folded_groups(const float* dw_acc, int32_t nparts, int32_t ntap, int32_t G, int32_t CoP, int32_t CiP, int32_t Cout, int32_t Cin,
                                             int64_t sCo, int64_t sCi, const float* coef, int32_t coef_gstride, int32_t coef_pitch,
                                             const double* cls_sums, int32_t padD, int32_t padH, int32_t padW, const float* w, float* dw,
                                             float* dbias_grad, double* bn_sums, int32_t bn_nrep, int32_t bn_cp, sp_stream_t stream) {
  SP_CHECK_ARG((ntap == 27 || (ntap == 1 && padD == 0 && padH == 0 && padW == 0)), "sp_wgrad_finish_folded_groups: 27 taps, or one (pointwise, padding 0)");
  SP_CHECK_ARG(dw_acc && coef && cls_sums && w && dw && G >= 1 && nparts >= G && nparts % G == 0 && G <= 16 && CoP >= Cout && CiP >= Cin && CiP >= 8,
               "sp_wgrad_finish_folded_groups: %d partial blocks for %d groups, tiles %d x %d", nparts, G, CoP, CiP);
  SP_CHECK_ARG(padD >= 0 && padD <= 2 && padH >= 0 && padH <= 2 && padW >= 0 && padW <= 2 && coef_pitch >= Cin && coef_gstride >= 3 * coef_pitch,
               "sp_wgrad_finish_folded_groups: padding 0..2, coefficient rows (scale, -, shift)");
  SP_CHECK_ARG(!bn_sums || (bn_nrep >= 1 && bn_cp >= Cin), "sp_wgrad_finish_folded_groups: bn_sums rows");
  const int64_t total = (int64_t)ntap * CoP * CiP;
  hipLaunchKernelGGL(wgrad_finish_folded_groups_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dw_acc,
                     nparts, ntap, G, CoP, CiP, Cout, Cin, sCo, sCi, coef, coef_gstride, coef_pitch, cls_sums, padD, padH, padW, w, dw, dbias_grad, bn_sums,
                     bn_nrep, bn_cp);
  SP_CHECK_LAUNCH("sp_wgrad_finish_folded_groups");
  return SP_OK;
}

extern "C" int sp_wgrad_finish_folded(float* dw_acc, int32_t nparts, const int32_t* tapsrc, int32_t ntap, int32_t CoP, int32_t CiP,
                                      int32_t Cout, int32_t Cin, int64_t sCo, int64_t sCi, const float* scale,
                                      const float* shift, const double* dbias_sums, float* dw, float* dbias_grad,
                                      const float* w_for_bn, double* bn_sums, int32_t bn_nrep, int32_t bn_cp, int32_t dbias_stride, sp_stream_t stream) {
  return wgrad_finish_folded_impl(dw_acc, nparts, tapsrc, ntap, CoP, CiP, Cout, Cin, sCo, sCi, scale, shift, dbias_sums, dw, dbias_grad,
                                  w_for_bn, bn_sums, bn_nrep, bn_cp, dbias_stride, 1.f, stream);
}
extern "C" int sp_wgrad_finish_folded_scaled(float* dw_acc, int32_t nparts, const int32_t* tapsrc, int32_t ntap, int32_t CoP, int32_t CiP,
                                             int32_t Cout, int32_t Cin, int64_t sCo, int64_t sCi, const float* scale,
                                             const float* shift, const double* dbias_sums, float* dw, float* dbias_grad,
                                             const float* w_for_bn, double* bn_sums, int32_t bn_nrep, int32_t bn_cp, int32_t dbias_stride,
                                             float acc_scale, sp_stream_t stream) {
  SP_CHECK_ARG(acc_scale > 0.f, "sp_wgrad_finish_folded_scaled: acc_scale");
  return wgrad_finish_folded_impl(dw_acc, nparts, tapsrc, ntap, CoP, CiP, Cout, Cin, sCo, sCi, scale, shift, dbias_sums, dw, dbias_grad,
                                  w_for_bn, bn_sums, bn_nrep, bn_cp, dbias_stride, acc_scale, stream);
}
