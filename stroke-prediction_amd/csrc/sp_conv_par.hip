// Parity classes of a transposed / strided-gradient convolution in ONE pass over the output (bf16).
//
// Replaces nn.ConvTranspose3d forward (Cae3D.py:178-204: kernel 2 / 3, stride 2) and the data gradient of the strided
// nn.Conv3d layers (Cae3D.py:45-64).  Such an op splits into up to eight sub-convolutions, one per output parity class
// (runtime/plan.py:_transposed_subs): class r writes the output voxels o = s*q + r and reads 1..8 taps of the input around q.
// Run one class per launch (sp_conv3d_igemm), every launch writes 32 bytes out of every 64 of its output rows and stages the
// same input tile again: measured 1.2 TB/s on the CAE's 24->16 @14x62x62 -> 28x124x124 gradient.  Here a workgroup owns a tile
// of the CLASS grid (16 rows of 16 voxels q) and walks all classes for it, so the 2x2x2 output voxels of every q leave together
// (whole lines in L2) and the input is fetched from HBM once.
//
// No LDS staging: the operands are small next to the output (the output is 8x the input voxels), so the activation fragment of a
// K step -- 16 bytes (eight channels) of voxel q + tap per lane -- is gathered straight from global memory (L1 / L2 serve the
// overlap between taps, rows and classes), and so are the weight fragments (1 KiB per (step, output tile), shared by every
// wave of the launch).  Out-of-volume taps read a zero page.  The K order of a class and its packed weights are the tiled
// kernel's (plan.SubConv.kmap): `gtab` holds, per K slot, the byte offset of (tap, octet) from the lane's base voxel and the
// tap's per-axis offsets for the bounds test.
//
// Measured (tools/probes/par_probe.py): 16->24 gradient of the CAE 196 -> 113 us (+ BatchNorm-backward sums 299 -> 137), the 16->16
// 2x2x2 transposed layer 144 -> 69 us (3.6 TB/s); the five class ops of a CAE step 946 -> 470 us.  An LDS-staged variant (tile and
// all weight fragments in one LDS-DMA burst) was 10-30 % faster on single ops and 0.1 ms SLOWER in the step, where these ops run
// beside the weight gradients of the side stream (57 KiB of LDS per workgroup); writing the x classes as dense half rows changed
// nothing either (the interleaved 32-byte pieces merge in L2); issuing ALL loads of a class (up to 8 steps) before its first MFMA,
// with 2 rows per wave and 4 or 8 waves, was 5-60 % slower than this one-step-ahead loop (fewer waves per CU).  None is kept.
//
// Epilogue as the tiled kernel's: bias, activation, 16-bit store, statistics of the stored values -- plain (sum, sum of squares:
// the next layer's BatchNorm) or stats_mode 1 (sum g, sum g*x with x = a.aux read at the same position: the BatchNorm backward of
// the layer whose data gradient this is) -- into the rows of the sample's BatchNorm group (group_batch).
#include "sp_common.h"

#define SP_PAR_MAXCLS 8
struct zm_dummy_t { uint32_t x, y; };      // four 16-bit values

struct ConvParCls {
  const bf16x8* w;              // [steps][NTtot][64] fragments
  int32_t steps, gofs;          // K steps of the class, its first slot in gtab
  int32_t Do, Ho, Wo;           // class grid
  int32_t ooD, ooH, ooW;        // output offset of the class
  int32_t o0D, o0H, o0W;        // input origin: i = q * stride + o0 + tap offset
  int32_t pad_;
};

struct ConvParDev {
  sp_conv_args a;               // the fields all classes share
  ConvParCls c[SP_PAR_MAXCLS];
  const int2* gtab;             // per K slot: (byte offset of (tap, octet) from the base voxel, oz | (4 + oy) << 8 | (8 + ox) << 16 | octet << 24)
  const void* zeros;
  int32_t ncls, cpb, ngtab, TD, TH;
  uint32_t ntx, nty, ntz, nblk;
  FastDiv d_tx, d_ty, d_tz, d_th;
};

template <int NT, int MT>
__global__ __launch_bounds__(256, 2) void conv_par_kernel(const ConvParDev P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_conv_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lv = lane & 15, lg = lane >> 4;
  int2* gt = reinterpret_cast<int2*>(lds);
  for (int i = tid; i < P.ngtab; i += 256) gt[i] = P.gtab[i];
  __syncthreads();

  uint32_t t = xcd_remap(blockIdx.x, P.nblk);
  uint32_t q = fdiv(t, P.d_tx); const int tx = t - q * P.ntx; t = q;
  q = fdiv(t, P.d_ty); const int ty = t - q * P.nty; t = q;
  q = fdiv(t, P.d_tz); const int tz = t - q * P.ntz; const int b = q;
  const int nt0 = blockIdx.y * NT;
  const int qx = tx * 16 + lv;
  int qz[MT], qy[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const uint32_t r = wave * MT + m;
    const uint32_t rz = fdiv(r, P.d_th);
    qz[m] = tz * P.TD + (int)rz;
    qy[m] = ty * P.TH + (int)(r - rz * P.TH);
  }
  const unsigned char* xin = reinterpret_cast<const unsigned char*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * a.CPi * 2;
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(P.zeros);
  bf16_t* __restrict__ yout = reinterpret_cast<bf16_t*>(a.y) + (size_t)b * a.YD * a.YH * a.YW * a.CPo;
  const bf16_t* __restrict__ auxin = reinterpret_cast<const bf16_t*>(a.aux) + (size_t)b * a.YD * a.YH * a.YW * a.CPo;
  const bool want_stats = a.stats != nullptr;

  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[n][j] = s2[n][j] = 0.f;

  const int c_begin = blockIdx.z * P.cpb;
  const int c_end = min(c_begin + P.cpb, P.ncls);
  for (int cls = c_begin; cls < c_end; ++cls) {
    const ConvParCls& C = P.c[cls];
    // base voxel of every row and which tap offsets (0..2 per axis) stay inside the input
    int xoff[MT];
    uint32_t vm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int iz = qz[m] * a.sD + C.o0D, iy = qy[m] * a.sH + C.o0H, ix = qx * a.sW + C.o0W;
      uint32_t v = 0;
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        if ((unsigned)(iz + o) < (unsigned)a.Di) v |= 1u << o;
        if ((unsigned)(iy + o) < (unsigned)a.Hi) v |= 16u << o;
        if ((unsigned)(ix + o) < (unsigned)a.Wi) v |= 256u << o;
      }
      vm[m] = v;
      xoff[m] = ((iz * a.Hi + iy) * a.Wi + ix) * a.CPi * 2;
    }
    // output positions of the class; BatchNorm-backward sums: the layer input there, requested before the K loop
    int obase[MT];
    zm_dummy_t xaux[NT][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bool valid = qz[m] < C.Do && qy[m] < C.Ho && qx < C.Wo;
      obase[m] = valid ? (((qz[m] * a.osD + C.ooD) * a.YH + (qy[m] * a.osH + C.ooH)) * a.YW + (qx * a.osW + C.ooW)) * a.CPo : -1;
    }
    if (want_stats && a.stats_mode == 1) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int c0 = (nt0 + n) * 16 + lg * 4;
          const bf16_t* ap = (obase[m] >= 0 && c0 < a.CPo) ? auxin + (size_t)obase[m] + c0 : reinterpret_cast<const bf16_t*>(zsrc);
          xaux[n][m] = *reinterpret_cast<const zm_dummy_t*>(ap);
        }
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int2* g = gt + C.gofs + lg;
    const bf16x8* wp = C.w + (size_t)nt0 * 64 + lane;
    const size_t fstride = (size_t)a.NTtot * 64;
    const int nst = C.steps;
    bf16x8 x0[MT], x1[MT], w0[NT], w1[NT];
#define PAR_LD(xd, wd, s_)                                                                                      \
  {                                                                                                             \
    const int2 e_ = g[(s_) * 4];                                                                                \
    const uint32_t sz_ = e_.y & 31, sy_ = (e_.y >> 8) & 31, sx_ = (e_.y >> 16) & 31;                            \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                            \
      const bool ok_ = ((vm[m] >> sz_) & (vm[m] >> sy_) & (vm[m] >> sx_) & 1u) != 0;                            \
      const unsigned char* p_ = ok_ ? xin + (int64_t)(xoff[m] + e_.x) : zsrc;                                   \
      xd[m] = *reinterpret_cast<const bf16x8*>(p_);                                                             \
    }                                                                                                           \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) wd[n] = wp[(size_t)(s_) * fstride + (size_t)n * 64];          \
  }
#define PAR_MMA(xv, wv)                                                                                         \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[n][m] = SP_MFMA16(wv[n], xv[m], acc[n][m], 0, 0, 0);
    PAR_LD(x0, w0, 0)
    for (int s = 0; s < nst; s += 2) {
      if (s + 1 < nst) PAR_LD(x1, w1, s + 1)
      PAR_MMA(x0, w0)
      if (s + 1 < nst) {
        if (s + 2 < nst) PAR_LD(x0, w0, s + 2)
        PAR_MMA(x1, w1)
      }
    }
#undef PAR_LD
#undef PAR_MMA

    // ---- epilogue of the class
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int c0 = (nt0 + n) * 16 + lg * 4;
      float bj[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) { const float4 bb = *reinterpret_cast<const float4*>(a.bias + c0); bj[0] = bb.x; bj[1] = bb.y; bj[2] = bb.z; bj[3] = bb.w; }
      const bool cok = c0 < a.CPo;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = act_fwd(a.act, a.act_param, acc[n][m][j] + bj[j]);
          v[j] = (c0 + j < a.Cout) ? z : 0.f;
        }
        if (obase[m] >= 0 && cok) {
          Store<bf16_t>::st4(yout + (size_t)obase[m] + c0, v);
          if (want_stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = bf2f(f2bf(v[j]));      // statistics of what is stored
            if (a.stats_mode == 1) {
              const float xv[4] = {sp_h2f_lo(xaux[n][m].x), sp_h2f_hi(xaux[n][m].x), sp_h2f_lo(xaux[n][m].y), sp_h2f_hi(xaux[n][m].y)};
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
  }

  if (want_stats) {
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
    double* const g_stats = a.stats + (size_t)(a.group_batch > 0 ? b / a.group_batch : 0) * a.stats_nrep * a.CPo * 2;
    for (int i = tid; i < NT * 16 * 2; i += 256) {
      const int c = nt0 * 16 + (i >> 1);
      if (c < a.CPo) atomicAdd(&g_stats[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (i & 1)], (double)sp_cols_sum(red, NT * 32, 4, i));
    }
  }
}


template <int NT, int MT>
static int launch_par(const ConvParDev& P, dim3 grid, int lds_bytes, hipStream_t st) {
  auto kern = conv_par_kernel<NT, MT>;
  SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_par");
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, P);
  SP_CHECK_LAUNCH("sp_conv3d_par");
  return SP_OK;
}

// args[0..n): the classes as sp_conv3d_igemm would take them (same x / y / bias / statistics / geometry of the whole op, the
// class's wfrag_hi, steps = ngroups * steps_per_group, Do / Ho / Wo, oo*, o0*); gtab: device table of all classes, gofs[i] the
// first slot of class i (gofs[n] = number of slots); zeros: >= 16 readable zero bytes.
extern "C" int sp_conv3d_par(const sp_conv_args* args, int32_t n, const void* gtab, const int32_t* gofs, const void* zeros, sp_stream_t stream) {
  SP_CHECK_ARG(args && n >= 1 && n <= SP_PAR_MAXCLS && gtab && gofs && zeros, "sp_conv3d_par: bad arguments (1..%d classes)", SP_PAR_MAXCLS);
  const sp_conv_args* a = &args[0];
  SP_CHECK_ARG(a->x && a->y, "sp_conv3d_par: null tensor");
  SP_CHECK_ARG(a->dtype_in == SP_BF16 && a->dtype_out == SP_BF16 && a->in_scale == nullptr && a->x_plane == 0 && !a->y8 && a->nslices <= 1,
               "sp_conv3d_par: channels-last bf16 in and out, no affine on load");
  SP_CHECK_ARG(a->CPi % 8 == 0 && a->CPo % 8 == 0 && a->NTtot >= 1 && a->NTtot * 16 >= a->Cout, "sp_conv3d_par: channel pitches (CPi %d, CPo %d, Cout %d, NTtot %d)", a->CPi, a->CPo, a->Cout, a->NTtot);
  SP_CHECK_ARG(!a->stats || (a->stats_nrep >= 1 && (a->stats_nrep & (a->stats_nrep - 1)) == 0), "sp_conv3d_par: stats_nrep must be a power of two");
  SP_CHECK_ARG(a->stats_mode == 0 || (a->stats_mode == 1 && a->stats && a->aux), "sp_conv3d_par: stats_mode 1 needs statistics rows and aux");
  SP_CHECK_ARG(a->group_batch == 0 || (a->group_batch > 0 && a->B % a->group_batch == 0), "sp_conv3d_par: group_batch %d does not divide the batch %d", a->group_batch, a->B);
  SP_CHECK_ARG((uint64_t)a->Di * a->Hi * a->Wi * a->CPi * 2 < (1ull << 31) && (uint64_t)a->YD * a->YH * a->YW * a->CPo < (1ull << 31), "sp_conv3d_par: sample too large for 32-bit offsets");
  ConvParDev P;
  P.a = *a;
  int mD = 0, mH = 0, mW = 0;
  for (int i = 0; i < n; ++i) {
    const sp_conv_args& c = args[i];
    SP_CHECK_ARG(c.wfrag_hi && c.Do > 0 && c.Ho > 0 && c.Wo > 0 && gofs[i + 1] - gofs[i] == 4 * c.ngroups * c.steps_per_group && gofs[i] >= 0,
                 "sp_conv3d_par: class %d: fragments, extents and %d K slots for %d steps", i, gofs[i + 1] - gofs[i], c.ngroups * c.steps_per_group);
    SP_CHECK_ARG(c.x == a->x && c.y == a->y && c.NTtot == a->NTtot && c.osD == a->osD && c.osH == a->osH && c.osW == a->osW && c.sD == a->sD && c.sH == a->sH && c.sW == a->sW,
                 "sp_conv3d_par: class %d differs from class 0 in tensors, tiles or strides", i);
    // every output voxel of the class lies inside y
    SP_CHECK_ARG((c.Do - 1) * c.osD + c.ooD < a->YD && (c.Ho - 1) * c.osH + c.ooH < a->YH && (c.Wo - 1) * c.osW + c.ooW < a->YW && c.ooD >= 0 && c.ooH >= 0 && c.ooW >= 0,
                 "sp_conv3d_par: class %d writes outside the output", i);
    ConvParCls& d = P.c[i];
    d.w = reinterpret_cast<const bf16x8*>(c.wfrag_hi);
    d.steps = c.ngroups * c.steps_per_group;
    d.gofs = gofs[i];
    d.Do = c.Do; d.Ho = c.Ho; d.Wo = c.Wo;
    d.ooD = c.ooD; d.ooH = c.ooH; d.ooW = c.ooW;
    d.o0D = c.o0D; d.o0H = c.o0H; d.o0W = c.o0W;
    d.pad_ = 0;
    mD = c.Do > mD ? c.Do : mD; mH = c.Ho > mH ? c.Ho : mH; mW = c.Wo > mW ? c.Wo : mW;
  }
  for (int i = n; i < SP_PAR_MAXCLS; ++i) P.c[i] = P.c[0];
  P.gtab = reinterpret_cast<const int2*>(gtab);
  P.zeros = zeros;
  P.ncls = n;
  P.ngtab = gofs[n];
  SP_CHECK_ARG(P.ngtab > 0 && (size_t)P.ngtab * 8 <= 60 * 1024, "sp_conv3d_par: K table of %d slots", P.ngtab);
  constexpr int MT = 4;
  P.TH = mH >= 8 ? 8 : 4;
  P.TD = 4 * MT / P.TH;
  P.d_th = make_fastdiv(P.TH);
  P.ntx = (mW + 15) / 16;
  P.nty = (mH + P.TH - 1) / P.TH;
  P.ntz = (mD + P.TD - 1) / P.TD;
  P.d_tx = make_fastdiv(P.ntx);
  P.d_ty = make_fastdiv(P.nty);
  P.d_tz = make_fastdiv(P.ntz);
  const uint64_t nblk = (uint64_t)P.ntx * P.nty * P.ntz * a->B;
  SP_CHECK_ARG(nblk < (1ull << 31), "sp_conv3d_par: grid too large");
  P.nblk = (uint32_t)nblk;
  const int NT = a->NTtot % 2 == 0 ? 2 : 1;
  // few tiles (the layers around the latent): one class per workgroup, so that the launch still fills the chip
  P.cpb = (nblk * (uint64_t)(a->NTtot / NT) >= 512 || n == 1) ? n : 1;
  dim3 grid(P.nblk, a->NTtot / NT, (n + P.cpb - 1) / P.cpb);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int lds_bytes = P.ngtab * 8;
  if (lds_bytes < 4 * NT * 32 * 4) lds_bytes = 4 * NT * 32 * 4;
  if (NT == 2) return launch_par<2, MT>(P, grid, lds_bytes, st);
  return launch_par<1, MT>(P, grid, lds_bytes, st);
}
