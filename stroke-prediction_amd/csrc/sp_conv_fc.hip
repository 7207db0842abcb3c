// Split-K convolution for FC-like layers: few output voxels, very deep K (the 800 <-> 100 channel layers around the CAE's
// latent, Cae3D.py:72-76 and 178-180: 1x10x10 <-> 3x12x12 voxels per sample, K = 27 x 800 = 21 600).  The tiled implicit-GEMM
// kernels give such a layer a handful of workgroups that each stage the whole 800-channel input through LDS in a dozen
// stage -> barrier -> compute rounds (270 us for 7.5 GFLOP); here
//   * K is split by TAP: grid = (tiles of 64 output voxels, blocks of 8 output tiles, taps) -- 700+ workgroups; a wave keeps
//     four K steps of operands in flight in a register ring (the plain loop waited out an L2 round trip per step);
//   * no LDS: a lane loads its B operand (8 channels of one input voxel, 16 bytes, channels-last) straight from global
//     memory -- out-of-volume taps load nothing -- and the weight fragments (A operand, packed by sp_conv_prep_weights in
//     tap-major K order) come from L2;
//   * a BatchNorm in front of the layer is applied in registers (x * scale + shift inside the volume, 0 outside: zero
//     padding follows the normalisation, Cae3D.py:41);
//   * every tap writes its fp32 partial tile; sp_conv_fc's second kernel adds the partials, applies bias / activation,
//     accumulates the statistics the next BatchNorm (or the BatchNorm backward of a data gradient) needs and stores.
#include "sp_common.h"

#define FC_NTB 8            // output tiles (of 16 channels) per workgroup

struct ConvFcDev {
  sp_conv_fc_args a;
  int32_t M, spt, octs, NTtot;       // output voxels, K steps per tap, input octets, output tiles
  FastDiv d_w, d_h, d_d;
};

#define FC_PD 4             // K steps in flight per wave (register ring): the loop is latency-bound without it
__global__ __launch_bounds__(256) void conv_fc_partial_kernel(const ConvFcDev P) {
  const sp_conv_fc_args& a = P.a;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vl = lane & 15, g = lane >> 4;
  const int tap = blockIdx.z;
  const int nt0 = blockIdx.y * FC_NTB;
  const int m = (blockIdx.x * 4 + wave) * 16 + vl;
  // output voxel -> input voxel of this tap
  uint32_t t = (uint32_t)(m < P.M ? m : P.M - 1);
  uint32_t q = fdiv(t, P.d_w); const int ox = t - q * a.Wo; t = q;
  q = fdiv(t, P.d_h); const int oy = t - q * a.Ho; t = q;
  q = fdiv(t, P.d_d); const int oz = t - q * a.Do; const int b = q;
  const int32_t* tp = a.taps + tap * 3;
  const int iz = oz * a.sD + a.o0D + tp[0], iy = oy * a.sH + a.o0H + tp[1], ix = ox * a.sW + a.o0W + tp[2];
  const bool inside = m < P.M && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
  // channels-last: 16 bytes of octet o at voxel*CPi + 8 o; plane-major ([CPi/16][B][D][H][W][16], a.x_plane elements per plane):
  // at (o >> 1) * x_plane + voxel*16 + 8 (o & 1)
  const int64_t vox = (((int64_t)b * a.Di + iz) * a.Hi + iy) * a.Wi + ix;
  const bf16_t* xp = reinterpret_cast<const bf16_t*>(a.x) + vox * (a.x_plane ? 16 : a.CPi);
  const uint4* wf = reinterpret_cast<const uint4*>(a.wfrag) + ((size_t)tap * P.spt * P.NTtot + nt0) * 64 + lane;
  f32x4 acc[FC_NTB];
#pragma unroll
  for (int n = 0; n < FC_NTB; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nn = min(FC_NTB, P.NTtot - nt0);
  // a wave whose 16 voxels all miss the volume for this tap has nothing to add (wave-uniform test)
  if (__builtin_amdgcn_ballot_w64(inside) != 0) {
    uint4 bq[FC_PD], aq[FC_PD][FC_NTB];
    auto fetch = [&](int s, int u) {
      const int oct = s * 4 + g;
      bq[u] = make_uint4(0, 0, 0, 0);
      if (inside && oct < P.octs)
        bq[u] = *reinterpret_cast<const uint4*>(a.x_plane ? xp + (int64_t)(oct >> 1) * a.x_plane + (oct & 1) * 8 : xp + oct * 8);
      const uint4* ws = wf + (size_t)s * P.NTtot * 64;
#pragma unroll
      for (int n = 0; n < FC_NTB; ++n) aq[u][n] = ws[(n < nn ? n : nn - 1) * 64];      // tiles past the last one: a valid fragment, result dropped
    };
    // No guards around the MFMAs (guards make hipcc shuttle accumulators between register files): spt is a multiple of
    // FC_PD (the plan pads the K steps of a tap with zero-weight octets), all FC_NTB tiles are computed, and the
    // fetch of a step past the end re-reads the last one.
#pragma unroll
    for (int u = 0; u < FC_PD; ++u) fetch(u, u);
    for (int s0 = 0; s0 < P.spt; s0 += FC_PD) {
#pragma unroll
      for (int u = 0; u < FC_PD; ++u) {
        const int s = s0 + u;
        {
          uint4 raw = bq[u];
          const int oct = s * 4 + g;
          if (a.in_scale && inside && oct < P.octs) {
            const float4 s0v = *reinterpret_cast<const float4*>(a.in_scale + oct * 8), s1v = *reinterpret_cast<const float4*>(a.in_scale + oct * 8 + 4);
            const float4 h0v = *reinterpret_cast<const float4*>(a.in_shift + oct * 8), h1v = *reinterpret_cast<const float4*>(a.in_shift + oct * 8 + 4);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            const float sc[8] = {s0v.x, s0v.y, s0v.z, s0v.w, s1v.x, s1v.y, s1v.z, s1v.w};
            const float sh[8] = {h0v.x, h0v.y, h0v.z, h0v.w, h1v.x, h1v.y, h1v.z, h1v.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float lo = sp_h2f_lo(w[j]), hi = sp_h2f_hi(w[j]);
              o[j] = sp_pack_bf16x2(fmaf(lo, sc[2 * j], sh[2 * j]), fmaf(hi, sc[2 * j + 1], sh[2 * j + 1]));
            }
            raw = make_uint4(o[0], o[1], o[2], o[3]);
          }
          const bf16x8 bfr = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
          for (int n = 0; n < FC_NTB; ++n)
            acc[n] = SP_MFMA16(__builtin_bit_cast(bf16x8, aq[u][n]), bfr, acc[n], 0, 0, 0);
          fetch(min(s + FC_PD, P.spt - 1), u);
        }
      }
    }
  }
  // D[cout = 4 g + j][voxel = vl]: 16-byte store per tile into this tap's partial block [M][NTtot*16]
  if (m < P.M) {
    float* pr = a.partial + ((size_t)tap * P.M + m) * (P.NTtot * 16) + nt0 * 16 + g * 4;
#pragma unroll
    for (int n = 0; n < FC_NTB; ++n)
      if (n < nn) *reinterpret_cast<float4*>(pr + n * 16) = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
  }
}

// one thread = one output voxel x 8 channels: sum the tap partials, bias, activation, statistics, store
template <typename TOUT>
__global__ __launch_bounds__(256) void conv_fc_finish_kernel(const ConvFcDev P) {
  const sp_conv_fc_args& a = P.a;
  __shared__ float s_red[256 * 16];
  const int OC = a.CPo / 8;
  const int pos = threadIdx.x / OC, oc = threadIdx.x - pos * OC;
  const int vpb = 256 / OC;
  const bool active = pos < vpb;
  const int NP = P.NTtot * 16;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  // BatchNorm groups: grid.y = group, its voxels [mlo, mhi), its statistics rows
  const int Mg = a.group_batch > 0 ? a.group_batch * a.Do * a.Ho * a.Wo : P.M;
  const int mlo = (int)blockIdx.y * Mg, mhi = min(P.M, mlo + Mg);
  if (active) {
    for (int m = mlo + blockIdx.x * vpb + pos; m < mhi; m += gridDim.x * vpb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (oc * 8 < NP) {
        const float* pr0 = a.partial + (size_t)m * NP + oc * 8;
        const size_t pstep = (size_t)P.M * NP;
        int tp = 0;
        for (; tp + 8 <= a.ntap; tp += 8) {          // eight independent loads in flight
          float4 p0[8], p1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            p0[u] = *reinterpret_cast<const float4*>(pr0 + (tp + u) * pstep);
            p1[u] = *reinterpret_cast<const float4*>(pr0 + (tp + u) * pstep + 4);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v[0] += p0[u].x; v[1] += p0[u].y; v[2] += p0[u].z; v[3] += p0[u].w;
            v[4] += p1[u].x; v[5] += p1[u].y; v[6] += p1[u].z; v[7] += p1[u].w;
          }
        }
        for (; tp < a.ntap; ++tp) {
          const float4 p0 = *reinterpret_cast<const float4*>(pr0 + tp * pstep), p1 = *reinterpret_cast<const float4*>(pr0 + tp * pstep + 4);
          v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = oc * 8 + j;
        float z = c < a.Cout ? v[j] + (a.bias ? a.bias[c] : 0.f) : 0.f;
        if (c < a.Cout) z = act_fwd(a.act, a.act_param, z);
        v[j] = z;
      }
      if (a.stats) {
        float x8[8];
        if (a.stats_mode == 1) Store<bf16_t>::ld8(reinterpret_cast<const bf16_t*>(a.aux) + (size_t)m * a.CPo + oc * 8, x8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += a.stats_mode == 1 ? v[j] * x8[j] : v[j] * v[j]; }
      }
      Store<TOUT>::st8(reinterpret_cast<TOUT*>(a.y) + (size_t)m * a.CPo + oc * 8, v);
    }
  }
  if (a.stats) {
    float4* d = reinterpret_cast<float4*>(s_red + threadIdx.x * 16);
    d[0] = make_float4(s1[0], s2[0], s1[1], s2[1]); d[1] = make_float4(s1[2], s2[2], s1[3], s2[3]);
    d[2] = make_float4(s1[4], s2[4], s1[5], s2[5]); d[3] = make_float4(s1[6], s2[6], s1[7], s2[7]);
    __syncthreads();
    // columns: (channel, {sum, second sum}); rows: the vpb voxel slots of this workgroup
    const int ncol = a.CPo * 2;
    double* o = a.stats + ((size_t)blockIdx.y * a.stats_nrep + (blockIdx.x & (a.stats_nrep - 1))) * ncol;
    for (int col = threadIdx.x; col < ncol; col += 256) {
      float tsum = 0.f;
      for (int k = 0; k < vpb; ++k) tsum += s_red[(k * OC) * 16 + col];
      atomicAdd(&o[col], (double)tsum);
    }
  }
}

// ---- pointwise (1x1x1) convolutions with few input channels (the CAE's tail, Cae3D.py:214-218: 16 -> 16 and 16 -> 1 at
// 28 x 128 x 128, and their data gradients): the same operand-from-global MFMA formulation with ONE K step and no K split
// -- weights resident in registers, a wave walks 16-voxel tiles, bias / activation / statistics / store in the same kernel.
// Through the tiled implicit-GEMM kernels such a layer cost 41-72 us for 117 MB of traffic.
#define PW_SPT 2            // K steps (32 channels each) at most: Cin <= 64
template <int SPT, int NTB, typename TOUT>      // compile-time K steps and output tiles per workgroup: no guard around an MFMA, no idle MFMA
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvFcDev P) {
  const sp_conv_fc_args& a = P.a;
  __shared__ float red[4 * NTB * 16 * 2];      // [wave][column]: added up in wave order (sp_cols_sum)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vl = lane & 15, g = lane >> 4;
  const int nt0 = blockIdx.y * NTB;
  const int nn = min(NTB, P.NTtot - nt0);
  const int32_t* tp = a.taps;
  const uint4* wf = reinterpret_cast<const uint4*>(a.wfrag) + (size_t)nt0 * 64 + lane;
  uint4 aq[SPT][NTB];
  float sc[SPT][8], sh[SPT][8];
  // BatchNorm groups: grid.z = group, its voxels [mlo, mhi), its scale / shift rows and statistics rows
  const int grp = blockIdx.z;
  const int Mg = a.group_batch > 0 ? a.group_batch * a.Do * a.Ho * a.Wo : P.M;
  const int mlo = grp * Mg, mhi = min(P.M, mlo + Mg);
#pragma unroll
  for (int s = 0; s < SPT; ++s) {
#pragma unroll
    for (int n = 0; n < NTB; ++n) aq[s][n] = wf[((size_t)s * P.NTtot + (n < nn ? n : nn - 1)) * 64];
    const int oct = s * 4 + g;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = a.in_scale && oct < P.octs;
      sc[s][j] = ok ? a.in_scale[(size_t)grp * a.coef_gstride + oct * 8 + j] : 1.f;
      sh[s][j] = ok ? a.in_shift[(size_t)grp * a.coef_gstride + oct * 8 + j] : 0.f;
    }
  }
  float bj[NTB][4], s1[NTB][4], s2[NTB][4];
#pragma unroll
  for (int n = 0; n < NTB; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = (nt0 + n) * 16 + g * 4 + j;
      bj[n][j] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
      s1[n][j] = s2[n][j] = 0.f;
    }
  const int ntile = (mhi - mlo + 15) / 16;
  constexpr int TPW = 4;             // tiles in flight per wave: the loads of four tiles are issued before the first is used (eight: no change)
  for (int base = (blockIdx.x * 4 + wave) * TPW; base < ntile; base += gridDim.x * 4 * TPW) {
    uint4 raw[TPW][SPT];
    int mm[TPW];
    bool ins[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int m = mlo + (base + u) * 16 + vl;
      mm[u] = m < mhi ? m : P.M;              // (>= P.M: outside -- the tile tail of a group must not reach into the next one)
      uint32_t t = (uint32_t)(m < mhi ? m : mhi - 1);
      uint32_t q = fdiv(t, P.d_w); const int ox = t - q * a.Wo; t = q;
      q = fdiv(t, P.d_h); const int oy = t - q * a.Ho; t = q;
      q = fdiv(t, P.d_d); const int oz = t - q * a.Do; const int b = q;
      const int iz = oz * a.sD + a.o0D + tp[0], iy = oy * a.sH + a.o0H + tp[1], ix = ox * a.sW + a.o0W + tp[2];
      ins[u] = m < mhi && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
      const int64_t vox = (((int64_t)b * a.Di + iz) * a.Hi + iy) * a.Wi + ix;
      const bf16_t* xp = reinterpret_cast<const bf16_t*>(a.x) + vox * (a.x_plane ? 16 : a.CPi);
#pragma unroll
      for (int s = 0; s < SPT; ++s) {
        const int oct = s * 4 + g;
        raw[u][s] = make_uint4(0, 0, 0, 0);
        if (ins[u] && oct < P.octs)
          raw[u][s] = *reinterpret_cast<const uint4*>(a.x_plane ? xp + (int64_t)(oct >> 1) * a.x_plane + (oct & 1) * 8 : xp + oct * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int m = mm[u];
      f32x4 acc[NTB];
#pragma unroll
      for (int n = 0; n < NTB; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SPT; ++s) {
        const int oct = s * 4 + g;
        uint4 r4 = raw[u][s];
        if (a.in_scale && ins[u] && oct < P.octs) {
          const uint32_t w[4] = {r4.x, r4.y, r4.z, r4.w};
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = sp_h2f_lo(w[j]), hi = sp_h2f_hi(w[j]);
            o[j] = sp_pack_bf16x2(fmaf(lo, sc[s][2 * j], sh[s][2 * j]), fmaf(hi, sc[s][2 * j + 1], sh[s][2 * j + 1]));
          }
          r4 = make_uint4(o[0], o[1], o[2], o[3]);
        }
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, r4);
#pragma unroll
        for (int n = 0; n < NTB; ++n)
          acc[n] = SP_MFMA16(__builtin_bit_cast(bf16x8, aq[s][n]), bfr, acc[n], 0, 0, 0);
      }
      // D[cout = 4 g + j][voxel = vl]
      if (m < P.M) {
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
          if (n < nn) {
            const int c0 = (nt0 + n) * 16 + g * 4;
            float v[4], x4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = (c0 + j < a.Cout) ? act_fwd(a.act, a.act_param, acc[n][j] + bj[n][j]) : 0.f;
              if (sizeof(TOUT) == 2) v[j] = bf2f(f2bf(v[j]));        // statistics of what is stored, like the tiled kernels
            }
            if (c0 < a.CPo) {
              if (a.stats) {
                if (a.stats_mode == 1) Store<bf16_t>::ld4(reinterpret_cast<const bf16_t*>(a.aux) + (size_t)m * a.CPo + c0, x4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1[n][j] += v[j]; s2[n][j] += a.stats_mode == 1 ? v[j] * x4[j] : v[j] * v[j]; }
              }
              Store<TOUT>::st4(reinterpret_cast<TOUT*>(a.y) + (size_t)m * a.CPo + c0, v);
            }
          }
        }
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int n = 0; n < NTB; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x1 = row16_sum(s1[n][j]), x2 = row16_sum(s2[n][j]);
        if (vl == 0 && n < nn) { red[wave * (NTB * 32) + (n * 16 + g * 4 + j) * 2] = x1; red[wave * (NTB * 32) + (n * 16 + g * 4 + j) * 2 + 1] = x2; }
      }
    __syncthreads();
    for (int i = threadIdx.x; i < nn * 32; i += 256) {
      const int c = nt0 * 16 + (i >> 1);
      if (c < a.CPo) atomicAdd(&a.stats[((size_t)grp * a.stats_nrep + (blockIdx.x & (a.stats_nrep - 1))) * a.CPo * 2 + (size_t)c * 2 + (i & 1)], (double)sp_cols_sum(red, NTB * 32, 4, i));
    }
  }
}

extern "C" int sp_conv_fc_workspace(int32_t B, int32_t Do, int32_t Ho, int32_t Wo, int32_t Cout, int32_t ntap, int64_t* floats) {
  SP_CHECK_ARG(floats && B >= 1 && Do >= 1 && Ho >= 1 && Wo >= 1 && Cout >= 1 && ntap >= 1, "sp_conv_fc_workspace: bad arguments");
  *floats = (int64_t)ntap * B * Do * Ho * Wo * ((Cout + 15) / 16 * 16);
  return SP_OK;
}

extern "C" int sp_conv_fc(const sp_conv_fc_args* a, sp_stream_t stream) {
  SP_CHECK_ARG(a && a->x && a->y && a->wfrag && a->taps, "sp_conv_fc: null pointer");
  SP_CHECK_ARG(a->partial || (a->ntap == 1 && a->CPi <= PW_SPT * 32), "sp_conv_fc: a partial buffer, or one tap and <= 64 input channels (pointwise mode)");
  SP_CHECK_ARG(a->x_plane == 0 || a->CPi % 16 == 0, "sp_conv_fc: plane-major input needs whole 16-channel planes");
  SP_CHECK_ARG(a->CPi % 8 == 0 && a->CPo % 8 == 0 && a->CPo <= 2048 && a->Cout <= a->CPo && a->Cout >= 1, "sp_conv_fc: channel pitches");
  SP_CHECK_ARG((a->in_scale == nullptr) == (a->in_shift == nullptr), "sp_conv_fc: scale and shift come together");
  SP_CHECK_ARG(a->ntap >= 1 && a->ntap <= 65535, "sp_conv_fc: taps");
  SP_CHECK_ARG(a->dtype_out == SP_BF16 || a->dtype_out == SP_F32, "sp_conv_fc: output type");
  SP_CHECK_ARG(!a->stats || (a->stats_nrep >= 1 && (a->stats_nrep & (a->stats_nrep - 1)) == 0), "sp_conv_fc: stats_nrep must be a power of two");
  SP_CHECK_ARG(!a->stats || a->stats_mode == 0 || (a->stats_mode == 1 && a->aux && a->dtype_out == SP_BF16), "sp_conv_fc: statistics mode");
  const int64_t M = (int64_t)a->B * a->Do * a->Ho * a->Wo;
  SP_CHECK_ARG(M >= 1 && M < (a->partial ? (1ll << 24) : (1ll << 31) - 16), "sp_conv_fc: output volume");
  SP_CHECK_ARG(a->group_batch >= 0 && (a->group_batch == 0 || (a->B % a->group_batch == 0 && a->coef_gstride >= 0)), "sp_conv_fc: group_batch %d must divide the batch %d", a->group_batch, a->B);
  const unsigned G = a->group_batch > 0 ? (unsigned)(a->B / a->group_batch) : 1u;
  ConvFcDev P;
  P.a = *a;
  P.M = (int32_t)M;
  P.octs = a->CPi / 8;
  if (!a->partial) {      // pointwise mode: one tap, K steps not padded (plan.py:fc_plan), everything in one kernel
    P.spt = (P.octs + 3) / 4;
    P.NTtot = (a->Cout + 15) / 16;
    P.d_w = make_fastdiv(a->Wo); P.d_h = make_fastdiv(a->Ho); P.d_d = make_fastdiv(a->Do);
    const int64_t tiles = (M / G + 15) / 16;
    unsigned gx = (unsigned)((tiles + 15) / 16 < 2048 / G ? (tiles + 15) / 16 : 2048 / G);
    if (gx < 1) gx = 1;
    const int ntb = P.NTtot >= 4 ? 4 : (P.NTtot >= 2 ? 2 : 1);
    dim3 grid(gx, (unsigned)((P.NTtot + ntb - 1) / ntb), G);
    hipStream_t st0 = reinterpret_cast<hipStream_t>(stream);
#define PW_CASE(S_, N_)                                                                                              \
  if (P.spt == S_ && ntb == N_) {                                                                                    \
    if (a->dtype_out == SP_BF16) hipLaunchKernelGGL((conv_pw_kernel<S_, N_, bf16_t>), grid, dim3(256), 0, st0, P);   \
    else hipLaunchKernelGGL((conv_pw_kernel<S_, N_, float>), grid, dim3(256), 0, st0, P);                            \
    SP_CHECK_LAUNCH("sp_conv_fc(pointwise)");                                                                        \
    return SP_OK;                                                                                                    \
  }
    PW_CASE(1, 1) PW_CASE(1, 2) PW_CASE(1, 4) PW_CASE(2, 1) PW_CASE(2, 2) PW_CASE(2, 4)
#undef PW_CASE
    sp_set_error("sp_conv_fc(pointwise): no kernel for %d K steps", P.spt);
    return SP_EINVAL;
  }
  P.spt = ((P.octs + 3) / 4 + FC_PD - 1) / FC_PD * FC_PD;          // runtime/plan.py:fc_plan pads the same way
  P.NTtot = (a->Cout + 15) / 16;
  P.d_w = make_fastdiv(a->Wo); P.d_h = make_fastdiv(a->Ho); P.d_d = make_fastdiv(a->Do);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((P.NTtot + FC_NTB - 1) / FC_NTB), (unsigned)a->ntap);
  hipLaunchKernelGGL(conv_fc_partial_kernel, grid, dim3(256), 0, st, P);
  SP_CHECK_LAUNCH("sp_conv_fc(partial)");
  const int vpb = 256 / (a->CPo / 8) > 0 ? 256 / (a->CPo / 8) : 1;
  SP_CHECK_ARG(a->CPo / 8 <= 256, "sp_conv_fc: too many output channels");
  unsigned fg = (unsigned)((M / G + vpb - 1) / vpb);
  if (fg > 1024 / G) fg = 1024 / G;
  if (fg < 1) fg = 1;
  if (a->dtype_out == SP_BF16) hipLaunchKernelGGL(conv_fc_finish_kernel<bf16_t>, dim3(fg, G), dim3(256), 0, st, P);
  else hipLaunchKernelGGL(conv_fc_finish_kernel<float>, dim3(fg, G), dim3(256), 0, st, P);
  SP_CHECK_LAUNCH("sp_conv_fc(finish)");
  return SP_OK;
}
