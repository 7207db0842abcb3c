// First layer of the U-Net (Unet3D.py:18-20 of block1: BatchNorm3d(2) -> Conv3d(2, 16, 3) -> LeakyReLU) read straight
// from the network input.  The input has TWO channels (CBV and TTD maps, NCDHW fp32): pushed through the generic
// 16-channel-plane kernels it is 8x zero padding in HBM, LDS and MFMA K (fwd 232 us, wgrad 359 us, layout copy 81 us,
// statistics 82 us per step at 4 x 2 x 128^3).  Here the im2col K dimension is packed instead:
//   K = 9 (dz,dy) groups x 8 = (4 dx x 2 channels), dx = 3 carrying zero weights  ->  3 MFMA 16x16x32 per 16 voxels,
// the staged tile is one dword per voxel (both channels as bf16), and nothing but the fp32 NCDHW input is read.
//   sp_bn_stats_ncdhw   batch statistics of the input for the first BatchNorm
//   sp_first_prep       conv weights with the BatchNorm folded in -> MFMA A fragments, folded bias
//   sp_first_conv_fwd   y = act(conv(W', x) + b'), channels-last bf16, plus the next BatchNorm's (sum, sum^2)
//   sp_first_wgrad      raw-input weight gradient as per-workgroup partial blocks [27][16][2] for
//                       sp_wgrad_finish_folded (which also yields the BatchNorm-backward sums: no data gradient)
#include "sp_common.h"

#define ST(s) reinterpret_cast<hipStream_t>(s)

namespace {
constexpr int FT_TZ = 2, FT_TY = 4, FT_TX = 64;          // output voxels per tile
constexpr int FT_XZ = FT_TZ + 2, FT_XY = FT_TY + 2;      // staged input rows
constexpr int FT_XP = 72;                                 // staged row pitch in voxels (>= TX + 2 + read-ahead)
constexpr int FT_ROWS = FT_XZ * FT_XY;

struct FirstDev {
  const float* x;                                         // [B][2][D][H][W] fp32
  int B, D, H, W, Do, Ho, Wo;
  int ntz, nty, ntx;
  uint32_t ntiles;
  FastDiv d_tx, d_ty, d_tz, d_xp, d_xy;
};

__device__ __forceinline__ void first_decode(const FirstDev& P, uint32_t tile, int& b, int& oz0, int& oy0, int& ox0) {
  const uint32_t q1 = fdiv(tile, P.d_tx); const int tx = tile - q1 * P.ntx;
  const uint32_t q2 = fdiv(q1, P.d_ty); const int ty = q1 - q2 * P.nty;
  const uint32_t q3 = fdiv(q2, P.d_tz); const int tz = q2 - q3 * P.ntz;
  b = q3; oz0 = tz * FT_TZ; oy0 = ty * FT_TY; ox0 = tx * FT_TX;
}

// stage the (TZ+2) x (TY+2) x XP input window of a tile; coordinates clamped (clamped voxels only ever meet
// zero weights / zero-filled dz).  INTERLEAVED: one dword per voxel = (c0 | c1 << 16) bf16.  Planar: [c][row][x] bf16.
template <bool INTERLEAVED>
__device__ __forceinline__ void first_stage_x(const FirstDev& P, int b, int oz0, int oy0, int ox0, uint32_t* xt) {
  const size_t plane = (size_t)P.D * P.H * P.W;
  const float* x0 = P.x + (size_t)b * 2 * plane;
  for (int i = threadIdx.x; i < FT_ROWS * FT_XP; i += 256) {
    const int row = fdiv(i, P.d_xp), xo = i - row * FT_XP;
    const int zz = fdiv(row, P.d_xy), yy = row - zz * FT_XY;
    const int gz = min(oz0 + zz, P.D - 1), gy = min(oy0 + yy, P.H - 1), gx = min(ox0 + xo, P.W - 1);
    const size_t o = ((size_t)gz * P.H + gy) * P.W + gx;
    const uint32_t v0 = f2bf(x0[o]), v1 = f2bf(x0[plane + o]);
    if (INTERLEAVED) {
      xt[i] = v0 | (v1 << 16);
    } else {
      bf16_t* pl = reinterpret_cast<bf16_t*>(xt);
      pl[i] = (bf16_t)v0;
      pl[FT_ROWS * FT_XP + i] = (bf16_t)v1;
    }
  }
}
// The same in two halves, so that the loads of the NEXT tile are in flight while the MFMAs of the current one run (the
// kernels are persistent loops over tiles; staged synchronously every tile waited out an HBM round trip of scattered
// 4-byte loads with nothing else to do): first_load_x issues the loads into registers, first_store_x converts and
// writes the LDS tile after the barrier that retires the previous tile.
__device__ __forceinline__ void raw8_to_f32(const uint4& r, float (&f)[8]) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = sp_h2f_lo(w[j]); f[2 * j + 1] = sp_h2f_hi(w[j]); }
}
constexpr int FT_XIT = (FT_ROWS * FT_XP + 255) / 256;      // elements per thread
__device__ __forceinline__ void first_load_x(const FirstDev& P, int b, int oz0, int oy0, int ox0, float (&r)[FT_XIT][2]) {
  const size_t plane = (size_t)P.D * P.H * P.W;
  const float* x0 = P.x + (size_t)b * 2 * plane;
#pragma unroll
  for (int it = 0; it < FT_XIT; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < FT_ROWS * FT_XP) {
      const int row = fdiv(i, P.d_xp), xo = i - row * FT_XP;
      const int zz = fdiv(row, P.d_xy), yy = row - zz * FT_XY;
      const int gz = min(oz0 + zz, P.D - 1), gy = min(oy0 + yy, P.H - 1), gx = min(ox0 + xo, P.W - 1);
      const size_t o = ((size_t)gz * P.H + gy) * P.W + gx;
      r[it][0] = x0[o]; r[it][1] = x0[plane + o];
    }
  }
}
template <bool INTERLEAVED, bool HL = false>
__device__ __forceinline__ void first_store_x(const float (&r)[FT_XIT][2], uint32_t* xt) {
#pragma unroll
  for (int it = 0; it < FT_XIT; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < FT_ROWS * FT_XP) {
      if (INTERLEAVED) {
        const uint32_t h = sp_pack_bf16x2(r[it][0], r[it][1]);
        xt[i] = h;
        if (HL) xt[FT_ROWS * FT_XP + i] = sp_pack_bf16x2(r[it][0] - sp_h2f_lo(h), r[it][1] - sp_h2f_hi(h));      // the lo tile behind the hi tile
      } else {
        bf16_t* pl = reinterpret_cast<bf16_t*>(xt);
        pl[i] = f2bf(r[it][0]);
        pl[FT_ROWS * FT_XP + i] = f2bf(r[it][1]);
      }
    }
  }
}
}  // namespace

// ------------------------------------------------------------------------------------------------ input statistics
// ROUND: values rounded to the 16-bit storage type first (the bf16 convolution consumes the rounded input); false: as they are
// (the bf16-pair first layer splits the fp32 input into hi + lo and consumes ~17 bits of it)
template <bool ROUND> __device__ __forceinline__ float first_rnd(float v) { return ROUND ? bf2f(f2bf(v)) : v; }
template <bool ROUND>
__global__ __launch_bounds__(256) void bn_stats_ncdhw_kernel(const float* __restrict__ x, int C, int64_t DHW, int CP,
                                                              double* __restrict__ sums, int nrep, int chunks) {
  // grid (chunks, B*C): one contiguous chunk of one (b, c) plane per workgroup
  const int bc = blockIdx.y, c = bc % C;
  const float* p = x + (size_t)bc * DHW;
  const int64_t per = (DHW + chunks - 1) / chunks, i0 = (int64_t)blockIdx.x * per, i1 = min(DHW, i0 + per);
  float s1 = 0.f, s2 = 0.f;
  if (((uintptr_t)(p + i0) & 15) == 0) {       // 16-byte loads, four per thread in flight (4-byte loads: 2.2 TB/s)
    const int64_t n4 = (i1 - i0) >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p + i0);
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
      float4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = p4[i + 256 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v0 = first_rnd<ROUND>(q[u].x), v1 = first_rnd<ROUND>(q[u].y), v2 = first_rnd<ROUND>(q[u].z), v3 = first_rnd<ROUND>(q[u].w);
        a1[u] += (v0 + v1) + (v2 + v3);
        a2[u] = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, a2[u]))));
      }
    }
    for (; i < n4; i += 256) {
      const float4 q = p4[i];
      const float v0 = first_rnd<ROUND>(q.x), v1 = first_rnd<ROUND>(q.y), v2 = first_rnd<ROUND>(q.z), v3 = first_rnd<ROUND>(q.w);
      a1[0] += (v0 + v1) + (v2 + v3);
      a2[0] = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, a2[0]))));
    }
    for (int64_t j = i0 + 4 * n4 + threadIdx.x; j < i1; j += 256) {
      const float v = first_rnd<ROUND>(p[j]);
      a1[1] += v; a2[1] = fmaf(v, v, a2[1]);
    }
    s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]); s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
      const float v = first_rnd<ROUND>(p[i]);
      s1 += v; s2 = fmaf(v, v, s2);
    }
  }
  __shared__ double r1[4], r2[4];
  const double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
  if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = d1; r2[threadIdx.x >> 6] = d2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* dst = sums + (size_t)((blockIdx.x + blockIdx.y) % nrep) * CP * 2 + c * 2;
    atomicAdd(dst, r1[0] + r1[1] + r1[2] + r1[3]);
    atomicAdd(dst + 1, r2[0] + r2[1] + r2[2] + r2[3]);
  }
}

static int bn_stats_ncdhw_impl(const float* x, int32_t B, int32_t C, int64_t DHW, int32_t CP, double* sums, int32_t nrep, bool round,
                               sp_stream_t stream) {
  SP_CHECK_ARG(x && sums && B >= 1 && C >= 1 && C <= CP && DHW >= 1 && nrep >= 1, "sp_bn_stats_ncdhw: bad arguments");
  int chunks = (int)((DHW + 8191) / 8192);          // ~2048 workgroups at B*C = 8, 128^3
  if (chunks > 1024) chunks = 1024;
  if (round) hipLaunchKernelGGL(bn_stats_ncdhw_kernel<true>, dim3(chunks, B * C), dim3(256), 0, ST(stream), x, C, DHW, CP, sums, nrep, chunks);
  else hipLaunchKernelGGL(bn_stats_ncdhw_kernel<false>, dim3(chunks, B * C), dim3(256), 0, ST(stream), x, C, DHW, CP, sums, nrep, chunks);
  SP_CHECK_LAUNCH("sp_bn_stats_ncdhw");
  return SP_OK;
}
extern "C" int sp_bn_stats_ncdhw(const float* x, int32_t B, int32_t C, int64_t DHW, int32_t CP, double* sums,
                                 int32_t nrep, sp_stream_t stream) {
  return bn_stats_ncdhw_impl(x, B, C, DHW, CP, sums, nrep, true, stream);
}
extern "C" int sp_bn_stats_ncdhw_f32(const float* x, int32_t B, int32_t C, int64_t DHW, int32_t CP, double* sums,
                                     int32_t nrep, sp_stream_t stream) {
  return bn_stats_ncdhw_impl(x, B, C, DHW, CP, sums, nrep, false, stream);
}

// ------------------------------------------------------------------------------------------------ weights
// A fragment of MFMA step s (0..2), lane l: row co = l % 16, K slice j = l / 16 -> group g = 4s + j = (dz, dy),
// elements e = 0..7 = (dx = e / 2, c = e % 2);  W' = W * scale[c],  b' = b + sum W * shift[c]
// bn.gamma != NULL (sp_first_prep_bn): scale / shift are finalized here from the batch (or running) statistics of the two input
// channels -- every workgroup derives them, workgroup 0 publishes them (sp_bn_fin_block) -- instead of by a launch of their own
__global__ void first_prep_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ scale,
                                  const float* __restrict__ shift, bf16_t* __restrict__ wfrag, float* __restrict__ bias_f, int Cout,
                                  bf16_t* __restrict__ wfrag_lo, const sp_bn_fin_args bn) {
  const int t = threadIdx.x;               // 192 threads = 3 steps x 64 lanes; blockIdx.x = output tile of 16 rows
  __shared__ float bn_sc[2][16];
  if (bn.gamma) {
    sp_bn_fin_block(bn, blockIdx.x == 0, bn_sc[0], bn_sc[1]);
    scale = bn_sc[0];
    shift = bn_sc[1];
  }
  // Cout = 32: tile t, row r holds channel (r / 4) * 8 + 4 t + r % 4 -- a lane of the forward kernel (rows 4 lg .. 4 lg + 3 of
  // both tiles) then owns EIGHT consecutive channels of its voxel and stores them with one 16-byte instruction
  const int s = t >> 6, l = t & 63, r_ = l & 15, g = 4 * s + (l >> 4);
  const int co = Cout == 32 ? (r_ >> 2) * 8 + 4 * (int)blockIdx.x + (r_ & 3) : r_;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int dx = e >> 1, c = e & 1;
    float v = 0.f;
    if (g < 9 && dx < 3 && co < Cout) v = w[(co * 2 + c) * 27 + g * 3 + dx] * (scale ? scale[c] : 1.f);
    const bf16_t hv = f2bf(v);
    wfrag[((size_t)blockIdx.x * 192 + t) * 8 + e] = hv;
    if (wfrag_lo) wfrag_lo[((size_t)blockIdx.x * 192 + t) * 8 + e] = f2bf(v - bf2f(hv));      // (bf16 pair: w = hi + lo)
  }
  if (t < Cout && blockIdx.x == 0) {      // (natural channel order; one thread per channel, its 54 loads independent)
    const int cb = t;
    float acc = b ? b[cb] : 0.f;
    if (shift) {
      float wv[54];
#pragma unroll
      for (int k = 0; k < 54; ++k) wv[k] = w[cb * 54 + k];
#pragma unroll
      for (int k = 0; k < 54; ++k) acc = fmaf(wv[k], shift[k / 27], acc);
    }
    bias_f[cb] = acc;
  }
}

// Cout = 16 (the 3-scale network of BASELINE.json configs[1]) or 32 (the 4-scale one, configs[4]): one or two output tiles
extern "C" int sp_first_supported(int32_t Cin, int32_t Cout, int32_t k) { return Cin == 2 && (Cout == 16 || Cout == 32) && k == 3; }

extern "C" int sp_first_prep_n(const float* w, const float* b, const float* scale, const float* shift, void* wfrag,
                               float* bias_f, int32_t Cout, sp_stream_t stream) {
  SP_CHECK_ARG(w && wfrag && bias_f && (Cout == 16 || Cout == 32), "sp_first_prep: null pointer / 16 or 32 output channels");
  hipLaunchKernelGGL(first_prep_kernel, dim3(Cout / 16), dim3(192), 0, ST(stream), w, b, scale, shift, (bf16_t*)wfrag, bias_f, Cout, (bf16_t*)nullptr, sp_bn_fin_args{});
  SP_CHECK_LAUNCH("sp_first_prep");
  return SP_OK;
}
extern "C" int sp_first_prep_hl(const float* w, const float* b, const float* scale, const float* shift, void* wfrag_hi, void* wfrag_lo,
                                float* bias_f, int32_t Cout, sp_stream_t stream) {
  SP_CHECK_ARG(w && wfrag_hi && wfrag_lo && bias_f && (Cout == 16 || Cout == 32), "sp_first_prep_hl: null pointer / 16 or 32 output channels");
  hipLaunchKernelGGL(first_prep_kernel, dim3(Cout / 16), dim3(192), 0, ST(stream), w, b, scale, shift, (bf16_t*)wfrag_hi, bias_f, Cout, (bf16_t*)wfrag_lo, sp_bn_fin_args{});
  SP_CHECK_LAUNCH("sp_first_prep_hl");
  return SP_OK;
}
extern "C" int sp_first_prep_bn(const float* w, const float* b, void* wfrag, void* wfrag_lo, float* bias_f, int32_t Cout,
                                const sp_bn_fin_args* bn, sp_stream_t stream) {
  SP_CHECK_ARG(w && wfrag && bias_f && bn && (Cout == 16 || Cout == 32), "sp_first_prep_bn: null pointer / 16 or 32 output channels");
  SP_CHECK_ARG(bn->gamma && bn->beta && bn->scale && bn->shift && bn->C == 2 && bn->CP >= 2 && bn->CP <= 16 && bn->nrep >= 1 &&
               (bn->training ? (bn->sums != nullptr && bn->count > 0) : (bn->running_mean && bn->running_var)),
               "sp_first_prep_bn: BatchNorm arguments (2 input channels, pitch <= 16)");
  hipLaunchKernelGGL(first_prep_kernel, dim3(Cout / 16), dim3(192), 0, ST(stream), w, b, (const float*)nullptr, (const float*)nullptr, (bf16_t*)wfrag,
                     bias_f, Cout, (bf16_t*)wfrag_lo, *bn);
  SP_CHECK_LAUNCH("sp_first_prep_bn");
  return SP_OK;
}
extern "C" int sp_first_prep(const float* w, const float* b, const float* scale, const float* shift, void* wfrag,
                             float* bias_f, sp_stream_t stream) {
  return sp_first_prep_n(w, b, scale, shift, wfrag, bias_f, 16, stream);
}

// ------------------------------------------------------------------------------------------------ forward
// NT output tiles of 16 channels (y rows of 16 NT channels); y8 != NULL: also the e4m3 plane-major copy of y
// ([NT][B][Do][Ho][Wo][16 bytes], the operand of an fp8 second layer -- csrc/sp_conv_zm8.hip), rounded from the stored value.
// HL: the bf16-pair form (SP_HL output, the "bf16x3" precision mode): the fp32 input is split into hi + lo bf16 tiles, the weights
// come as hi + lo fragments, every product is three MFMAs and y is written as a pair (y, y_lo); statistics of the fp32 values.
template <int NT, bool HL = false>
__global__ __launch_bounds__(256) void first_fwd_kernel(const FirstDev P, const bf16x8* __restrict__ wfrag,
                                                         const float* __restrict__ bias, int act, float ap,
                                                         bf16_t* __restrict__ y, double* __restrict__ stats, int nrep,
                                                         unsigned char* __restrict__ y8, int64_t y8_plane,
                                                         const bf16x8* __restrict__ wfrag_lo = nullptr, bf16_t* __restrict__ y_lo = nullptr) {
  __shared__ __attribute__((aligned(16))) uint32_t xt[(HL ? 2 : 1) * FT_ROWS * FT_XP];
  __shared__ float red[4 * 32 * NT];      // [wave][column]: added up in wave order (sp_cols_sum)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, n = lane & 15;
  bf16x8 af[NT][3], afl[HL ? NT : 1][3];
  int goff[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      af[nt][s] = wfrag[(nt * 3 + s) * 64 + lane];
      if (HL) afl[nt][s] = wfrag_lo[(nt * 3 + s) * 64 + lane];
    }
    const int g = min(4 * s + lg, 8);                       // groups 9..11 carry zero weights: any finite data will do
    goff[s] = ((g / 3) * FT_XY + (g % 3)) * FT_XP;
  }
  // output channel of (tile nt, row 4 lg + j): NT = 1: 4 lg + j; NT = 2: 8 lg + 4 nt + j (first_prep_kernel) -- this lane's eight
  // values of a voxel are consecutive channels
  constexpr int CQ = 4 * NT;                        // channels per lane and voxel
  float bj[NT][4], s1[NT][4], s2[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bj[nt][j] = bias[lg * CQ + nt * 4 + j]; s1[nt][j] = s2[nt][j] = 0.f; }
  const float slope = act == SP_ACT_LEAKY ? ap : 1.f;
  const bool lin = act == SP_ACT_LEAKY || act == SP_ACT_NONE;

  float xr[FT_XIT][2];
  if (blockIdx.x < P.ntiles) {
    int b, oz0, oy0, ox0;
    first_decode(P, blockIdx.x, b, oz0, oy0, ox0);
    first_load_x(P, b, oz0, oy0, ox0, xr);
  }
  for (uint32_t tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    int b, oz0, oy0, ox0;
    first_decode(P, tile, b, oz0, oy0, ox0);
    __syncthreads();                                        // the previous tile has been consumed
    first_store_x<true, HL>(xr, xt);
    __syncthreads();
    if (tile + gridDim.x < P.ntiles) {                      // next tile's input: in flight during this tile's MFMAs and stores
      int b2, oz2, oy2, ox2;
      first_decode(P, tile + gridDim.x, b2, oz2, oy2, ox2);
      first_load_x(P, b2, oz2, oy2, ox2, xr);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr, zz = row >> 2, yy = row & 3;
      const int oz = oz0 + zz, oy = oy0 + yy;
      const bool rok = oz < P.Do && oy < P.Ho;
      const int rbase = (zz * FT_XY + yy) * FT_XP + n;
      const size_t vrow = (((size_t)b * P.Do + oz) * P.Ho + oy) * P.Wo;      // first voxel of the output row
      bf16_t* yrow = y + vrow * (16 * NT) + lg * CQ;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        union { uint32_t u[4]; bf16x8 v; } bq[3], bql[HL ? 3 : 1];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const uint32_t* p = xt + goff[s] + rbase + t * 16;
          bq[s].u[0] = p[0]; bq[s].u[1] = p[1]; bq[s].u[2] = p[2]; bq[s].u[3] = p[3];
          if (HL) { const uint32_t* pl = p + FT_ROWS * FT_XP; bql[s].u[0] = pl[0]; bql[s].u[1] = pl[1]; bql[s].u[2] = pl[2]; bql[s].u[3] = pl[3]; }
        }
        const int ox = ox0 + t * 16 + n;
        float q[NT][4];
        uint32_t w2[NT][2], w2l[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            acc = SP_MFMA16(af[nt][s], bq[s].v, acc, 0, 0, 0);
            if (HL) {
              acc = SP_MFMA16(af[nt][s], bql[s].v, acc, 0, 0, 0);
              acc = SP_MFMA16(afl[nt][s], bq[s].v, acc, 0, 0, 0);
            }
          }
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float z = acc[j] + bj[nt][j];
            v[j] = lin ? fmaxf(z, slope * z) : act_fwd(act, ap, z);
          }
          if (HL) {
            sp_hl_split4(v, w2[nt][0], w2[nt][1], w2l[nt][0], w2l[nt][1]);
            q[nt][0] = v[0]; q[nt][1] = v[1]; q[nt][2] = v[2]; q[nt][3] = v[3];      // statistics of the fp32 values
          } else {
            w2[nt][0] = sp_pack_bf16x2(v[0], v[1]); w2[nt][1] = sp_pack_bf16x2(v[2], v[3]);
            q[nt][0] = sp_h2f_lo(w2[nt][0]); q[nt][1] = sp_h2f_hi(w2[nt][0]);      // statistics (and the e4m3 copy) of what is stored
            q[nt][2] = sp_h2f_lo(w2[nt][1]); q[nt][3] = sp_h2f_hi(w2[nt][1]);
          }
        }
        if (rok && ox < P.Wo) {
          if (y) {      // (NULL: fp8 mode, every reader takes the e4m3 copy)
            bf16_t* yp = yrow + (size_t)ox * (16 * NT);
            if (NT == 2) *reinterpret_cast<uint4*>(yp) = make_uint4(w2[0][0], w2[0][1], w2[NT - 1][0], w2[NT - 1][1]);
            else *reinterpret_cast<uint2*>(yp) = make_uint2(w2[0][0], w2[0][1]);
          }
          if (HL) {
            bf16_t* ypl = y_lo + vrow * (16 * NT) + lg * CQ + (size_t)ox * (16 * NT);
            if (NT == 2) *reinterpret_cast<uint4*>(ypl) = make_uint4(w2l[0][0], w2l[0][1], w2l[NT - 1][0], w2l[NT - 1][1]);
            else *reinterpret_cast<uint2*>(ypl) = make_uint2(w2l[0][0], w2l[0][1]);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[nt][j] += q[nt][j]; s2[nt][j] = fmaf(q[nt][j], q[nt][j], s2[nt][j]); }
          if (y8) {      // channels lg * CQ .. + CQ - 1: plane (lg * CQ) / 16, bytes (lg * CQ) % 16 .. of the voxel's 16
            int r8[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              r8[nt] = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(q[nt][0], -448.f, 448.f), __builtin_amdgcn_fmed3f(q[nt][1], -448.f, 448.f), 0, false);
              r8[nt] = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(q[nt][2], -448.f, 448.f), __builtin_amdgcn_fmed3f(q[nt][3], -448.f, 448.f), r8[nt], true);
            }
            unsigned char* p8 = y8 + (size_t)((lg * CQ) >> 4) * y8_plane + (vrow + ox) * 16 + ((lg * CQ) & 15);
            if (NT == 2) *reinterpret_cast<int2*>(p8) = make_int2(r8[0], r8[NT - 1]);
            else *reinterpret_cast<int*>(p8) = r8[0];
          }
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x1 = row16_sum(s1[nt][j]), x2 = row16_sum(s2[nt][j]);
        if (n == 0) { red[wave * (32 * NT) + (lg * CQ + nt * 4 + j) * 2] = x1; red[wave * (32 * NT) + (lg * CQ + nt * 4 + j) * 2 + 1] = x2; }
      }
    __syncthreads();
    if (tid < 32 * NT) atomicAdd(&stats[(size_t)(blockIdx.x % nrep) * (32 * NT) + tid], (double)sp_cols_sum(red, 32 * NT, 4, tid));
  }
}

static int first_geometry(FirstDev& P, const float* x, int B, int D, int H, int W) {
  P.x = x; P.B = B; P.D = D; P.H = H; P.W = W; P.Do = D - 2; P.Ho = H - 2; P.Wo = W - 2;
  P.ntz = (P.Do + FT_TZ - 1) / FT_TZ; P.nty = (P.Ho + FT_TY - 1) / FT_TY; P.ntx = (P.Wo + FT_TX - 1) / FT_TX;
  const uint64_t nt = (uint64_t)P.ntz * P.nty * P.ntx * B;
  if (nt >= (1ull << 31)) return -1;
  P.ntiles = (uint32_t)nt;
  P.d_tx = make_fastdiv(P.ntx); P.d_ty = make_fastdiv(P.nty); P.d_tz = make_fastdiv(P.ntz);
  P.d_xp = make_fastdiv(FT_XP); P.d_xy = make_fastdiv(FT_XY);
  return 0;
}

extern "C" int sp_first_conv_fwd_n(const float* x, int32_t B, int32_t D, int32_t H, int32_t W, const void* wfrag,
                                   const float* bias_f, int32_t act, float act_param, void* y, double* stats, int32_t nrep,
                                   int32_t Cout, void* y8, int64_t y8_plane, sp_stream_t stream) {
  SP_CHECK_ARG(x && wfrag && bias_f && (y || y8) && B >= 1 && D >= 3 && H >= 3 && W >= 3 && (Cout == 16 || Cout == 32), "sp_first_conv_fwd: bad arguments (y may be NULL when the e4m3 copy is asked for)");
  SP_CHECK_ARG(!stats || nrep >= 1, "sp_first_conv_fwd: stats replicas");
  SP_CHECK_ARG(!y8 || y8_plane >= (int64_t)B * (D - 2) * (H - 2) * (W - 2) * 16, "sp_first_conv_fwd: y8_plane smaller than a plane of the output");
  FirstDev P;
  SP_CHECK_ARG(first_geometry(P, x, B, D, H, W) == 0, "sp_first_conv_fwd: too many tiles");
  static const unsigned cap_ = getenv("SP_FIRST_BLOCKS") ? (unsigned)atoi(getenv("SP_FIRST_BLOCKS")) : 2048u;
  const unsigned grid = P.ntiles < cap_ ? P.ntiles : cap_;
  if (Cout == 16)
    hipLaunchKernelGGL((first_fwd_kernel<1, false>), dim3(grid), dim3(256), 0, ST(stream), P, (const bf16x8*)wfrag, bias_f, act, act_param,
                       (bf16_t*)y, stats, nrep, (unsigned char*)y8, y8_plane, (const bf16x8*)nullptr, (bf16_t*)nullptr);
  else
    hipLaunchKernelGGL((first_fwd_kernel<2, false>), dim3(grid), dim3(256), 0, ST(stream), P, (const bf16x8*)wfrag, bias_f, act, act_param,
                       (bf16_t*)y, stats, nrep, (unsigned char*)y8, y8_plane, (const bf16x8*)nullptr, (bf16_t*)nullptr);
  SP_CHECK_LAUNCH("sp_first_conv_fwd");
  return SP_OK;
}
// the bf16-pair form: y / y_lo = hi / lo halves of the output (SP_HL), weights as hi + lo fragments from sp_first_prep_hl
extern "C" int sp_first_conv_fwd_hl(const float* x, int32_t B, int32_t D, int32_t H, int32_t W, const void* wfrag_hi, const void* wfrag_lo,
                                    const float* bias_f, int32_t act, float act_param, void* y, void* y_lo, double* stats, int32_t nrep,
                                    int32_t Cout, sp_stream_t stream) {
  SP_CHECK_ARG(x && wfrag_hi && wfrag_lo && bias_f && y && y_lo && B >= 1 && D >= 3 && H >= 3 && W >= 3 && (Cout == 16 || Cout == 32), "sp_first_conv_fwd_hl: bad arguments");
  SP_CHECK_ARG(!stats || nrep >= 1, "sp_first_conv_fwd_hl: stats replicas");
  FirstDev P;
  SP_CHECK_ARG(first_geometry(P, x, B, D, H, W) == 0, "sp_first_conv_fwd_hl: too many tiles");
  static const unsigned cap_ = getenv("SP_FIRST_BLOCKS") ? (unsigned)atoi(getenv("SP_FIRST_BLOCKS")) : 2048u;
  const unsigned grid = P.ntiles < cap_ ? P.ntiles : cap_;
  if (Cout == 16)
    hipLaunchKernelGGL((first_fwd_kernel<1, true>), dim3(grid), dim3(256), 0, ST(stream), P, (const bf16x8*)wfrag_hi, bias_f, act, act_param,
                       (bf16_t*)y, stats, nrep, (unsigned char*)nullptr, (int64_t)0, (const bf16x8*)wfrag_lo, (bf16_t*)y_lo);
  else
    hipLaunchKernelGGL((first_fwd_kernel<2, true>), dim3(grid), dim3(256), 0, ST(stream), P, (const bf16x8*)wfrag_hi, bias_f, act, act_param,
                       (bf16_t*)y, stats, nrep, (unsigned char*)nullptr, (int64_t)0, (const bf16x8*)wfrag_lo, (bf16_t*)y_lo);
  SP_CHECK_LAUNCH("sp_first_conv_fwd_hl");
  return SP_OK;
}
extern "C" int sp_first_conv_fwd(const float* x, int32_t B, int32_t D, int32_t H, int32_t W, const void* wfrag,
                                 const float* bias_f, int32_t act, float act_param, void* y, double* stats, int32_t nrep,
                                 sp_stream_t stream) {
  return sp_first_conv_fwd_n(x, B, D, H, W, wfrag, bias_f, act, act_param, y, stats, nrep, 16, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][c][dz,dy,dx] = sum_v dz[v][co] * x[c][v + (dz,dy,dx)].  MFMA with K = 32 voxels along x:
//   A = dz^T (rows co), transposed out of the staged channels-last dz tile by ds_read_b64_tr_b16 (natural K order);
//   B = columns r = (dz,dy) (9 of 16 used), one column tile per channel c, eight consecutive x voxels per lane from
//       the planar staged input; the three dx taps are the same five dwords shifted by 0, 1 and 2 elements.
// FUSED: dz is not read but formed on the fly, dz = (c0*g + c1*y + c2) * act'(y) (BatchNorm backward of the NEXT layer
// and the activation derivative, i.e. sp_bn_act_bwd), and its per-channel sum is accumulated for the bias gradient --
// the 2 x 256 MB round trip of dz through HBM and one launch disappear (nobody else reads this layer's dz).
// Y8 (with FUSED): y comes as its e4m3 plane-major copy [NT][B][Do][Ho][Wo][16 bytes] (fp8 mode: the 16-bit tensor is not stored) --
// yg then points at that copy, y8_plane = bytes per plane.
template <bool FUSED, int ACT, int NT, bool Y8 = false>      // ACT >= 0: activation fixed at compile time (no per-element branch chain); NT output tiles
__global__ __launch_bounds__(256) void first_wgrad_kernel(const FirstDev P, const bf16_t* __restrict__ dzg,
                                                           const bf16_t* __restrict__ gg, const bf16_t* __restrict__ yg,
                                                           const float* __restrict__ coef, int act, float ap,
                                                           double* __restrict__ dbias, float* __restrict__ part, int64_t y8_plane = 0) {
  constexpr int CO = 16 * NT;                                                  // output channels (row pitch of dz / g / y)
  constexpr int DZP = FT_TZ * FT_TY * FT_TX * 32;                              // bytes of one 16-channel plane of the dz tile
  __shared__ __attribute__((aligned(16))) uint32_t xt[FT_ROWS * FT_XP];        // planar: [c][row][x] bf16
  __shared__ __attribute__((aligned(16))) unsigned char dzt[NT * DZP];         // [tile][row][x][16 ch]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, n = lane & 15;
  const int lq = n >> 2, lp = n & 3;
  const int r = n < 9 ? n : 0;                             // idle columns read tap 0 (their results are never flushed)
  const int boff = ((r / 3) * FT_XY + (r % 3)) * FT_XP + 8 * lg;        // elements, per channel plane
  const int aoff0 = (8 * lg + lq) * 32 + lp * 8, aoff1 = aoff0 + 4 * 32;   // bytes inside a 32-voxel K block
  f32x4 acc[NT][3][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int d = 0; d < 3; ++d) acc[t][d][0] = acc[t][d][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* xpl = reinterpret_cast<const bf16_t*>(xt);
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  // a thread always stages the same channel octet q = tid % (2 NT) (i = tid + 256 it, 256 % (2 NT) == 0)
  const int q8 = tid & (2 * NT - 1);
  float k0[8], k1[8], k2[8], dsum[8];
  if (FUSED) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = q8 * 8 + j;
      k0[j] = coef[c]; k1[j] = coef[CO + c]; k2[j] = coef[2 * CO + c]; dsum[j] = 0.f;
    }
  }

  // operands of a tile are fetched into registers one tile ahead (see first_load_x): x, and g / y (or dz)
  constexpr int DZIT = FT_TZ * FT_TY * FT_TX * 2 * NT / 256;
  float xr[FT_XIT][2];
  uint4 gq[DZIT], yq[DZIT];
  // chunk i = (row, x, octet): 16 bytes = 8 channels of one output voxel
  auto chunk = [&](int i, int& row, int& vx) { const int v = i / (2 * NT); vx = v & (FT_TX - 1); row = v >> 6; };
  static_assert(FT_TX == 64, "chunk decoding assumes 64-voxel tile rows");
  auto load_dz = [&](int b, int oz0, int oy0, int ox0) {
#pragma unroll
    for (int it = 0; it < DZIT; ++it) {
      int row, vx;
      chunk(tid + it * 256, row, vx);
      const int oz = oz0 + (row >> 2), oy = oy0 + (row & 3), ox = ox0 + vx;
      gq[it] = yq[it] = make_uint4(0, 0, 0, 0);
      if (oz < P.Do && oy < P.Ho && ox < P.Wo) {
        const size_t o = ((((size_t)b * P.Do + oz) * P.Ho + oy) * P.Wo + ox) * CO + q8 * 8;
        if (FUSED) {
          gq[it] = *reinterpret_cast<const uint4*>(gg + o);
          if constexpr (Y8) {
            const size_t v_ = (((size_t)b * P.Do + oz) * P.Ho + oy) * P.Wo + ox;
            const uint2 t8 = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(yg) + (size_t)(q8 >> 1) * y8_plane + v_ * 16 + (q8 & 1) * 8);
            yq[it] = make_uint4(t8.x, t8.y, 0, 0);
          } else yq[it] = *reinterpret_cast<const uint4*>(yg + o);
        } else gq[it] = *reinterpret_cast<const uint4*>(dzg + o);
      }
    }
  };
  if (blockIdx.x < P.ntiles) {
    int b, oz0, oy0, ox0;
    first_decode(P, blockIdx.x, b, oz0, oy0, ox0);
    first_load_x(P, b, oz0, oy0, ox0, xr);
    load_dz(b, oz0, oy0, ox0);
  }
  for (uint32_t tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    int b, oz0, oy0, ox0;
    first_decode(P, tile, b, oz0, oy0, ox0);
    __syncthreads();
    first_store_x<false>(xr, xt);
    // dz tile: per 16-channel tile a plane [row = zz*TY + yy][x][16 ch], zero where the output voxel does not exist
#pragma unroll
    for (int it = 0; it < DZIT; ++it) {
      int row, vx;
      chunk(tid + it * 256, row, vx);
      const int oz = oz0 + (row >> 2), oy = oy0 + (row & 3), ox = ox0 + vx;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (oz < P.Do && oy < P.Ho && ox < P.Wo) {
        if (FUSED) {
          float g8[8], y8[8], d8[8];
          raw8_to_f32(gq[it], g8);
          if constexpr (Y8) {
            typedef float f2v_ __attribute__((ext_vector_type(2)));
            const f2v_ a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yq[it].x, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yq[it].x, true);
            const f2v_ a2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yq[it].y, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)yq[it].y, true);
            y8[0] = a0[0]; y8[1] = a0[1]; y8[2] = a1[0]; y8[3] = a1[1]; y8[4] = a2[0]; y8[5] = a2[1]; y8[6] = a3[0]; y8[7] = a3[1];
          } else raw8_to_f32(yq[it], y8);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            d8[j] = (k0[j] * g8[j] + k1[j] * y8[j] + k2[j]) * act_bwd_from_y(ACT >= 0 ? ACT : act, ap, y8[j]);
            dsum[j] += d8[j];
          }
          uint32_t w4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w4[j] = sp_pack_bf16x2(d8[2 * j], d8[2 * j + 1]);
          v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        } else {
          v = gq[it];
        }
      }
      *reinterpret_cast<uint4*>(dzt + (q8 >> 1) * DZP + (size_t)((row * FT_TX + vx) * 2 + (q8 & 1)) * 16) = v;
    }
    __syncthreads();
    if (tile + gridDim.x < P.ntiles) {
      int b2, oz2, oy2, ox2;
      first_decode(P, tile + gridDim.x, b2, oz2, oy2, ox2);
      first_load_x(P, b2, oz2, oy2, ox2, xr);
      load_dz(b2, oz2, oy2, ox2);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr, zz = row >> 2, yy = row & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const unsigned char* ab = dzt + t * DZP + (row * FT_TX + ks * 32) * 32;
          const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ab + aoff0));
          const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ab + aoff1));
          af[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16_t* bp = xpl + c * (FT_ROWS * FT_XP) + (zz * FT_XY + yy) * FT_XP + ks * 32 + boff;
          const uint4 q = *reinterpret_cast<const uint4*>(bp);
          const uint32_t q4 = *reinterpret_cast<const uint32_t*>(bp + 8);
          union { uint32_t u[4]; bf16x8 v; } b0, b1, b2;
          b0.u[0] = q.x; b0.u[1] = q.y; b0.u[2] = q.z; b0.u[3] = q.w;
          b1.u[0] = __builtin_amdgcn_alignbit(q.y, q.x, 16); b1.u[1] = __builtin_amdgcn_alignbit(q.z, q.y, 16);
          b1.u[2] = __builtin_amdgcn_alignbit(q.w, q.z, 16); b1.u[3] = __builtin_amdgcn_alignbit(q4, q.w, 16);
          b2.u[0] = q.y; b2.u[1] = q.z; b2.u[2] = q.w; b2.u[3] = q4;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t][0][c] = SP_MFMA16(af[t], b0.v, acc[t][0][c], 0, 0, 0);
            acc[t][1][c] = SP_MFMA16(af[t], b1.v, acc[t][1][c], 0, 0, 0);
            acc[t][2][c] = SP_MFMA16(af[t], b2.v, acc[t][2][c], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- flush: D[row = co = lg*4 + j][col = r = n]; one partial block [tap = r*3 + dx][co][c] per workgroup, the four
  // waves' tiles added through LDS
  __syncthreads();
  float* st = reinterpret_cast<float*>(dzt);      // [4 waves][27 * CO * 2] (27.6 KiB at CO = 32; the dz tile holds 32 KiB)
  static_assert(4 * 27 * CO * 2 * 4 <= NT * DZP, "per-wave weight-gradient slabs do not fit the dz tile");
  if (n < 9) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) st[wave * (27 * CO * 2) + ((n * 3 + d) * CO + t * 16 + lg * 4 + j) * 2 + c] = acc[t][d][c][j];
  }
  __syncthreads();
  for (int i = tid; i < 27 * CO * 2; i += 256) part[(size_t)blockIdx.x * (27 * CO * 2) + i] = sp_cols_sum(st, 27 * CO * 2, 4, i);      // wave order
  if (FUSED && dbias) {       // sum of dz per output channel: threads with equal tid % (2 NT) hold the same eight channels
    __syncthreads();
    float* rd = st;           // [4 waves][CO]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = dsum[j];
      // lanes of one octet: xor-reduce over the other lane bits
#pragma unroll
      for (int o = 32; o >= 2 * NT; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane < 2 * NT) rd[wave * CO + lane * 8 + j] = v;
    }
    __syncthreads();
    if (tid < CO) atomicAdd(&dbias[(blockIdx.x % SP_REDUCE_ROWS) * CO + tid], (double)sp_cols_sum(rd, CO, 4, tid));
  }
}

extern "C" int sp_first_wgrad_n(const float* x, const void* dz, int32_t B, int32_t D, int32_t H, int32_t W, float* partials,
                                int32_t nblocks, int32_t Cout, sp_stream_t stream) {
  SP_CHECK_ARG(x && dz && partials && B >= 1 && D >= 3 && H >= 3 && W >= 3 && nblocks >= 1 && (Cout == 16 || Cout == 32), "sp_first_wgrad: bad arguments");
  FirstDev P;
  SP_CHECK_ARG(first_geometry(P, x, B, D, H, W) == 0, "sp_first_wgrad: too many tiles");
  if (Cout == 16)
    hipLaunchKernelGGL((first_wgrad_kernel<false, -1, 1>), dim3(nblocks), dim3(256), 0, ST(stream), P, (const bf16_t*)dz, nullptr, nullptr,
                       nullptr, 0, 0.f, nullptr, partials);
  else
    hipLaunchKernelGGL((first_wgrad_kernel<false, -1, 2>), dim3(nblocks), dim3(256), 0, ST(stream), P, (const bf16_t*)dz, nullptr, nullptr,
                       nullptr, 0, 0.f, nullptr, partials);
  SP_CHECK_LAUNCH("sp_first_wgrad");
  return SP_OK;
}
extern "C" int sp_first_wgrad(const float* x, const void* dz, int32_t B, int32_t D, int32_t H, int32_t W, float* partials,
                              int32_t nblocks, sp_stream_t stream) {
  return sp_first_wgrad_n(x, dz, B, D, H, W, partials, nblocks, 16, stream);
}

extern "C" int sp_first_wgrad_fused_n(const float* x, const void* g, const void* y, const float* coef, int32_t act, float act_param,
                                      int32_t B, int32_t D, int32_t H, int32_t W, float* partials, int32_t nblocks,
                                      double* dbias_sums, int32_t Cout, sp_stream_t stream) {
  SP_CHECK_ARG(x && g && y && coef && partials && dbias_sums && B >= 1 && D >= 3 && H >= 3 && W >= 3 && nblocks >= 1 && (Cout == 16 || Cout == 32),
               "sp_first_wgrad_fused: bad arguments");
  FirstDev P;
  SP_CHECK_ARG(first_geometry(P, x, B, D, H, W) == 0, "sp_first_wgrad_fused: too many tiles");
#define SP_FW(A_, N_)                                                                                                             \
  hipLaunchKernelGGL((first_wgrad_kernel<true, A_, N_>), dim3(nblocks), dim3(256), 0, ST(stream), P, nullptr, (const bf16_t*)g,    \
                     (const bf16_t*)y, coef, act, act_param, dbias_sums, partials)
  if (Cout == 16) { if (act == SP_ACT_LEAKY) SP_FW(SP_ACT_LEAKY, 1); else SP_FW(-1, 1); }
  else { if (act == SP_ACT_LEAKY) SP_FW(SP_ACT_LEAKY, 2); else SP_FW(-1, 2); }
#undef SP_FW
  SP_CHECK_LAUNCH("sp_first_wgrad_fused");
  return SP_OK;
}
// the same with y as its e4m3 plane-major copy (sp_first_conv_fwd_n's y8 / y8_plane: the fp8 mode stores no 16-bit y)
extern "C" int sp_first_wgrad_fused_y8(const float* x, const void* g, const void* y8, int64_t y8_plane, const float* coef, int32_t act,
                                       float act_param, int32_t B, int32_t D, int32_t H, int32_t W, float* partials, int32_t nblocks,
                                       double* dbias_sums, int32_t Cout, sp_stream_t stream) {
  SP_CHECK_ARG(x && g && y8 && coef && partials && dbias_sums && B >= 1 && D >= 3 && H >= 3 && W >= 3 && nblocks >= 1 && (Cout == 16 || Cout == 32),
               "sp_first_wgrad_fused_y8: bad arguments");
  SP_CHECK_ARG(y8_plane >= (int64_t)B * (D - 2) * (H - 2) * (W - 2) * 16, "sp_first_wgrad_fused_y8: y8_plane smaller than a plane of the output");
  FirstDev P;
  SP_CHECK_ARG(first_geometry(P, x, B, D, H, W) == 0, "sp_first_wgrad_fused_y8: too many tiles");
#define SP_FW(A_, N_)                                                                                                             \
  hipLaunchKernelGGL((first_wgrad_kernel<true, A_, N_, true>), dim3(nblocks), dim3(256), 0, ST(stream), P, nullptr, (const bf16_t*)g, \
                     (const bf16_t*)y8, coef, act, act_param, dbias_sums, partials, y8_plane)
  if (Cout == 16) { if (act == SP_ACT_LEAKY) SP_FW(SP_ACT_LEAKY, 1); else SP_FW(-1, 1); }
  else { if (act == SP_ACT_LEAKY) SP_FW(SP_ACT_LEAKY, 2); else SP_FW(-1, 2); }
#undef SP_FW
  SP_CHECK_LAUNCH("sp_first_wgrad_fused_y8");
  return SP_OK;
}
extern "C" int sp_first_wgrad_fused(const float* x, const void* g, const void* y, const float* coef, int32_t act, float act_param,
                                    int32_t B, int32_t D, int32_t H, int32_t W, float* partials, int32_t nblocks,
                                    double* dbias_sums, sp_stream_t stream) {
  return sp_first_wgrad_fused_n(x, g, y, coef, act, act_param, B, D, H, W, partials, nblocks, dbias_sums, 16, stream);
}
