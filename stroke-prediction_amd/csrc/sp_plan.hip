// Self-sufficient C entry points for the hot convolution: nn.Conv3d(Cin, Cout, 3, stride 1, padding 0) of the U-Net's
// Block3x3x3 (Unet3D.py:19,22) and its data gradient, bf16 channels-last, on the z-marching kernel (sp_conv_zm.hip).
// A C / C++ caller needs include/stroke_amd.h only: sp_conv3d_plan sizes the workspace from a POD descriptor,
// sp_conv3d_init builds the kernel's K tables on the host (the C twin of runtime/plan.py:zm_plan -- tests/test_cabi.py
// holds the two against each other) and uploads them, sp_conv3d_set_weights packs fp32 weights in nn.Conv3d layout (with
// an optional BatchNorm folded in), sp_conv3d_run launches.
#include <string.h>
#include <vector>
#include "sp_common.h"

#define SP_CHECK_HIP(call, what)                                                                   \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) { sp_set_error("%s: %s", what, hipGetErrorString(e_)); return SP_EHIP; } \
  } while (0)
#define SP_PLAN_ITW 18          // staged row: 16 output voxels + 2 (plan.py: ZM_ITW)

static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

extern "C" int sp_conv3d_plan(const sp_conv3d_desc* d, sp_conv3d_plan_t* p) {
  SP_CHECK_ARG(d && p, "sp_conv3d_plan: null pointer");
  SP_CHECK_ARG(d->B >= 1 && d->D >= 1 && d->H >= 1 && d->W >= 1, "sp_conv3d_plan: empty volume");
  SP_CHECK_ARG(d->Cin >= 16 && d->Cout >= 16 && d->Cin % 16 == 0 && d->Cout % 16 == 0, "sp_conv3d_plan: channels must be multiples of 16 (pad with zero channels)");
  SP_CHECK_ARG(d->padD >= 0 && d->padD <= 2 && d->padH >= 0 && d->padH <= 2 && d->padW >= 0 && d->padW <= 2, "sp_conv3d_plan: padding 0..2 per axis");
  SP_CHECK_ARG(!d->transposed || !d->grad, "sp_conv3d_plan: transposed = 1 describes the forward of nn.ConvTranspose3d (grad = 0)");
  const int pd[3] = {d->padD, d->padH, d->padW}, in[3] = {d->D, d->H, d->W};
  for (int ax = 0; ax < 3; ++ax)
    SP_CHECK_ARG(d->transposed ? in[ax] + 2 - 2 * pd[ax] >= 1 : in[ax] + 2 * pd[ax] >= 3, "sp_conv3d_plan: volume smaller than the kernel");
  memset(p, 0, sizeof(*p));
  // the op's own roles: the data gradient reads Cout-channel dz on the conv's output grid and writes Cin channels
  p->cin_op = d->grad ? d->Cout : d->Cin;
  p->cout_op = d->grad ? d->Cin : d->Cout;
  p->mirror = (d->grad || d->transposed) ? 1 : 0;
  p->P = p->cin_op / 16; p->NT = p->cout_op / 16;
  SP_CHECK_ARG(sp_conv3d_zm_config(p->P, p->NT, &p->MT, &p->NSLOT, &p->NW) == SP_OK,
               "sp_conv3d_plan: no z-marching kernel for %d input planes x %d output tiles (use sp_conv3d_igemm with a host plan)", p->P, p->NT);
  p->KS = (18 * p->P + 3) / 4;
  p->nsteps = 3 * p->KS;
  p->ITH = p->NW * p->MT + 2;
  if (d->transposed) {      // y = full correlation of x with the mirrored kernel, cropped by the padding: out = in + 2 - 2 pad
    p->Di = d->D; p->Hi = d->H; p->Wi = d->W;
    p->Do = d->D + 2 - 2 * d->padD; p->Ho = d->H + 2 - 2 * d->padH; p->Wo = d->W + 2 - 2 * d->padW;
    p->o0 = -(2 - d->padD); p->o0H = -(2 - d->padH); p->o0W = -(2 - d->padW);
  } else if (!d->grad) {
    p->Di = d->D; p->Hi = d->H; p->Wi = d->W;
    p->Do = d->D + 2 * d->padD - 2; p->Ho = d->H + 2 * d->padH - 2; p->Wo = d->W + 2 * d->padW - 2;
    p->o0 = -d->padD; p->o0H = -d->padH; p->o0W = -d->padW;
  } else {                  // data gradient of the (padded) convolution: dz on the conv's output grid in, the conv's input grid out
    p->Di = d->D + 2 * d->padD - 2; p->Hi = d->H + 2 * d->padH - 2; p->Wi = d->W + 2 * d->padW - 2;
    p->Do = d->D; p->Ho = d->H; p->Wo = d->W;
    p->o0 = -(2 - d->padD); p->o0H = -(2 - d->padH); p->o0W = -(2 - d->padW);
  }
  p->x_elems = (int64_t)d->B * p->Di * p->Hi * p->Wi * p->cin_op;
  p->y_elems = (int64_t)d->B * p->Do * p->Ho * p->Wo * p->cout_op;
  int64_t off = 0;
  p->off_zero = off;  off += 256;
  p->off_ktab = off;  off += align256((int64_t)p->KS * 4 * 4);
  p->off_kmap = off;  off += align256((int64_t)p->nsteps * 4 * 4);
  p->off_bias = off;  off += align256((int64_t)p->NT * 16 * 4);
  p->off_wfrag = off; off += align256((int64_t)p->nsteps * p->NT * 64 * 8 * 2);
  p->workspace_bytes = off;
  return SP_OK;
}

// ktab[4*KS], kmap[12*KS] in the kernel's K order (one input plane feeds the output planes dz = 0, 1, 2 above it; a step covers
// four of the 18 P in-plane octets (dy, dx, plane, half)).  Forward: source tap (dz, dy, dx) of the weight; data gradient:
// the mirrored tap, 26 - index ("full" correlation with the flipped kernel).
extern "C" int sp_conv3d_tables(const sp_conv3d_desc* d, const sp_conv3d_plan_t* p, int32_t* ktab, int32_t* kmap) {
  SP_CHECK_ARG(d && p && ktab && kmap, "sp_conv3d_tables: null pointer");
  const int P = p->P, KS = p->KS, n = 18 * P;
  for (int i = 0; i < 12 * KS; ++i) kmap[i] = -1;
  for (int e = 0; e < n; ++e) {
    const int t2d = e / (2 * P), rest = e % (2 * P), pl = rest / 2, o = rest % 2, dy = t2d / 3, dx = t2d % 3;
    ktab[e] = ((pl * p->ITH + dy) * SP_PLAN_ITW + dx) * 32 + o * 16;
    for (int dz = 0; dz < 3; ++dz) {
      const int tap = (dz * 3 + dy) * 3 + dx;
      kmap[dz * KS * 4 + e] = ((p->mirror ? 26 - tap : tap) << 16) | (pl * 2 + o);
    }
  }
  for (int e = n; e < 4 * KS; ++e) ktab[e] = ktab[e - 2];      // zero-weight padding octets: any valid, conflict-free address
  return SP_OK;
}

extern "C" int sp_conv3d_init(const sp_conv3d_desc* d, const sp_conv3d_plan_t* p, void* workspace, sp_stream_t stream) {
  SP_CHECK_ARG(d && p && workspace, "sp_conv3d_init: null pointer");
  std::vector<int32_t> ktab(4 * p->KS), kmap(12 * p->KS);
  int rc = sp_conv3d_tables(d, p, ktab.data(), kmap.data());
  if (rc != SP_OK) return rc;
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  SP_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)p->off_wfrag, st), "sp_conv3d_init");          // zero page, tables, bias
  SP_CHECK_HIP(hipMemcpyAsync(ws + p->off_ktab, ktab.data(), ktab.size() * 4, hipMemcpyHostToDevice, st), "sp_conv3d_init");
  SP_CHECK_HIP(hipMemcpyAsync(ws + p->off_kmap, kmap.data(), kmap.size() * 4, hipMemcpyHostToDevice, st), "sp_conv3d_init");
  SP_CHECK_HIP(hipStreamSynchronize(st), "sp_conv3d_init");                                   // the host tables die with this frame
  return SP_OK;
}

extern "C" int sp_conv3d_set_weights(const sp_conv3d_desc* d, const sp_conv3d_plan_t* p, void* workspace, const float* w,
                                     const float* bias, const float* bn_scale, const float* bn_shift, sp_stream_t stream) {
  SP_CHECK_ARG(d && p && workspace && w, "sp_conv3d_set_weights: null pointer");
  SP_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "sp_conv3d_set_weights: BatchNorm scale and shift come together");
  SP_CHECK_ARG(!d->grad || (!bias && !bn_scale), "sp_conv3d_set_weights: the data gradient takes the plain weights");
  SP_CHECK_ARG(!bn_scale || (!d->transposed && d->padD == 0 && d->padH == 0 && d->padW == 0),
               "sp_conv3d_set_weights: a BatchNorm folds into un-padded convolutions only (zero padding applies after the normalisation)");
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  const int32_t* kmap = reinterpret_cast<const int32_t*>(ws + p->off_kmap);
  void* hi = ws + p->off_wfrag;
  float* bias_out = reinterpret_cast<float*>(ws + p->off_bias);
  // element (co', ci', tap) of the op = w[co][ci][tap] of the nn.Conv3d weight [Cout][Cin][27]; roles swap for the gradient
  // (nn.ConvTranspose3d keeps [Cin][Cout][27]: the op's input channel is the slow index)
  const int64_t sCo = d->grad ? 27 : (d->transposed ? 27 : (int64_t)d->Cin * 27);
  const int64_t sCi = d->grad ? (int64_t)d->Cin * 27 : (d->transposed ? (int64_t)d->Cout * 27 : 27);
  if (bn_scale)
    return sp_conv_prep_folded(w, sCo, sCi, p->cout_op, p->cin_op, kmap, p->nsteps, p->NT, hi, nullptr, bn_scale, 27, bias, bn_shift,
                               bias_out, p->NT * 16, stream);
  int rc = sp_conv_prep_weights(w, sCo, sCi, p->cout_op, p->cin_op, kmap, p->nsteps, p->NT, hi, nullptr, nullptr, stream);
  if (rc != SP_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bias) { SP_CHECK_HIP(hipMemcpyAsync(bias_out, bias, (size_t)p->cout_op * 4, hipMemcpyDeviceToDevice, st), "sp_conv3d_set_weights"); }
  else { SP_CHECK_HIP(hipMemsetAsync(bias_out, 0, (size_t)p->NT * 16 * 4, st), "sp_conv3d_set_weights"); }
  return SP_OK;
}

extern "C" int sp_conv3d_run(const sp_conv3d_desc* d, const sp_conv3d_plan_t* p, const void* workspace, const void* x, void* y,
                             int32_t with_bias, int32_t act, float act_param, double* stats, int32_t stats_nrep, int64_t x_plane,
                             sp_stream_t stream) {
  SP_CHECK_ARG(d && p && workspace && x && y, "sp_conv3d_run: null pointer");
  const unsigned char* ws = static_cast<const unsigned char*>(workspace);
  sp_conv_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y;
  a.wfrag_hi = ws + p->off_wfrag;
  a.bias = with_bias ? reinterpret_cast<const float*>(ws + p->off_bias) : nullptr;
  a.stats = stats; a.stats_nrep = stats ? stats_nrep : 1;
  a.ktab = reinterpret_cast<const int32_t*>(ws + p->off_ktab);
  a.dtype_in = a.dtype_out = SP_BF16;
  a.B = d->B; a.Di = p->Di; a.Hi = p->Hi; a.Wi = p->Wi; a.CPi = p->cin_op;
  a.Do = a.YD = p->Do; a.Ho = a.YH = p->Ho; a.Wo = a.YW = p->Wo; a.CPo = p->cout_op;
  a.osD = a.osH = a.osW = 1;
  a.Cout = p->cout_op;
  a.sD = a.sH = a.sW = 1;
  const bool legacy = !d->transposed && d->padD == 0 && d->padH == 0 && d->padW == 0;      // (plans filled by an older caller: o0 only)
  a.o0D = p->o0; a.o0H = legacy ? p->o0 : p->o0H; a.o0W = legacy ? p->o0 : p->o0W;
  a.TD = 1; a.TH = p->NW * p->MT;
  a.ITD = 1; a.ITH = p->ITH; a.ITW = SP_PLAN_ITW;
  a.MT = p->MT; a.NT = a.NTtot = p->NT;
  a.ngroups = 1; a.octs_per_group = 2 * p->P; a.opp = 2; a.vsb = 32;
  a.steps_per_group = p->nsteps;
  a.act = act; a.act_param = act_param;
  a.dma = 1; a.persist = 5;
  a.x_plane = x_plane;
  return sp_conv3d_zm(&a, ws + p->off_zero, stream);
}

// ------------------------------------------------------------------------------------------------ weight gradient, header only
// dW of the same layer (Block3x3x3's conv, Unet3D.py:19,22) for a caller with this header and nothing else: the plan sizes one
// workspace (tap tables + one partial block [27][Cout][Cin] per persistent workgroup), init uploads the tables, run = the
// row-sliding LDS-DMA kernel (csrc/sp_wgrad_zr.hip, through sp_conv3d_wgrad) + the finish pass that adds the blocks up into dw --
// with the layer's input BatchNorm folded (dW = s * acc + t * sum dz; the BatchNorm-backward sums come out of the same pass) or
// without.  Mirrors ops.WgradRunner (the Python host side); tests/test_gpu_round2.py holds the two against each other.
extern "C" int sp_conv3d_wgrad_plan(const sp_conv3d_desc* d, sp_conv3d_wgrad_plan_t* p) {
  SP_CHECK_ARG(d && p, "sp_conv3d_wgrad_plan: null pointer");
  SP_CHECK_ARG(d->B >= 1 && d->D >= 3 && d->H >= 3 && d->W >= 3, "sp_conv3d_wgrad_plan: volume smaller than the kernel");
  SP_CHECK_ARG(d->Cin >= 16 && d->Cout >= 16 && d->Cin % 16 == 0 && d->Cout % 16 == 0, "sp_conv3d_wgrad_plan: channels must be multiples of 16 (pad with zero channels)");
  memset(p, 0, sizeof(*p));
  p->CoT = d->Cout / 16; p->CiT = d->Cin / 16;
  p->Do = d->D - 2; p->Ho = d->H - 2; p->Wo = d->W - 2;
  // one partial block per persistent workgroup: >= ~512 output voxels each, the (cout, cin) tile grid times the workgroups ~2 per CU
  const int cob = p->CoT >= 4 ? 4 : (p->CoT >= 2 ? 2 : 1);
  const int yz = ((p->CoT + cob - 1) / cob) * p->CiT;                   // (one input plane per workgroup: cib = 1)
  const int64_t vox = (int64_t)d->B * p->Do * p->Ho * p->Wo;
  int64_t nb = 512 / yz < vox / 512 ? 512 / yz : vox / 512;
  nb = (nb < 8 ? 8 : nb) / 8 * 8;
  p->nblocks = (int32_t)nb;
  int64_t off = 0;
  p->off_taps = off;   off += align256(27 * 3 * 4);
  p->off_tapsrc = off; off += align256(27 * 4);
  p->off_acc = off;    off += align256((int64_t)p->nblocks * 27 * d->Cout * d->Cin * 4);
  p->workspace_bytes = off;
  return SP_OK;
}

extern "C" int sp_conv3d_wgrad_init(const sp_conv3d_desc* d, const sp_conv3d_wgrad_plan_t* p, void* workspace, sp_stream_t stream) {
  SP_CHECK_ARG(d && p && workspace, "sp_conv3d_wgrad_init: null pointer");
  int32_t taps[27 * 3], src[27];
  for (int t = 0; t < 27; ++t) { taps[3 * t] = t / 9; taps[3 * t + 1] = (t / 3) % 3; taps[3 * t + 2] = t % 3; src[t] = t; }
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  SP_CHECK_HIP(hipMemcpyAsync(ws + p->off_taps, taps, sizeof taps, hipMemcpyHostToDevice, st), "sp_conv3d_wgrad_init");
  SP_CHECK_HIP(hipMemcpyAsync(ws + p->off_tapsrc, src, sizeof src, hipMemcpyHostToDevice, st), "sp_conv3d_wgrad_init");
  SP_CHECK_HIP(hipStreamSynchronize(st), "sp_conv3d_wgrad_init");
  return SP_OK;
}

extern "C" int sp_conv3d_wgrad_run(const sp_conv3d_desc* d, const sp_conv3d_wgrad_plan_t* p, void* workspace, const void* x, const void* dz,
                                   float* dw, const float* bn_scale, const float* bn_shift, const double* dbias_sums, int32_t dbias_stride,
                                   float* dbias_grad, const float* w_for_bn, double* bn_sums, int32_t bn_nrep, int64_t x_plane,
                                   sp_stream_t stream) {
  SP_CHECK_ARG(d && p && workspace && x && dz && dw, "sp_conv3d_wgrad_run: null pointer");
  SP_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "sp_conv3d_wgrad_run: BatchNorm scale and shift come together");
  SP_CHECK_ARG(!bn_scale || dbias_sums, "sp_conv3d_wgrad_run: the folded form needs dbias_sums = sum over voxels of dz per output channel");
  SP_CHECK_ARG(!bn_sums || (bn_scale && w_for_bn && bn_nrep >= 1), "sp_conv3d_wgrad_run: BatchNorm-backward sums need the folded form and the conv weight");
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  sp_wgrad_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dz = dz;
  a.dw_acc = reinterpret_cast<float*>(ws + p->off_acc);
  a.taps = reinterpret_cast<const int32_t*>(ws + p->off_taps);
  a.dtype = SP_BF16;
  a.B = d->B; a.Di = d->D; a.Hi = d->H; a.Wi = d->W; a.CPi = d->Cin;
  a.Do = p->Do; a.Ho = p->Ho; a.Wo = p->Wo; a.CPo = d->Cout;
  a.sD = a.sH = a.sW = 1;
  a.ntap = 27; a.kD = a.kH = a.kW = 3;
  a.CoT = p->CoT; a.CiT = p->CiT;
  a.nblocks = p->nblocks; a.dma = 1; a.parts = 1; a.cib = 1; a.zs = 1;
  a.x_plane = x_plane;
  int rc = sp_conv3d_wgrad(&a, stream);
  if (rc != SP_OK) return rc;
  const int32_t* tapsrc = reinterpret_cast<const int32_t*>(ws + p->off_tapsrc);
  const int64_t sCo = (int64_t)d->Cin * 27, sCi = 27;                  // nn.Conv3d weight layout [Cout][Cin][27]
  if (bn_scale)
    return sp_wgrad_finish_folded(a.dw_acc, p->nblocks, tapsrc, 27, d->Cout, d->Cin, d->Cout, d->Cin, sCo, sCi, bn_scale, bn_shift, dbias_sums,
                                  dw, dbias_grad, w_for_bn, bn_sums, bn_nrep, 0, dbias_stride, stream);
  return sp_wgrad_finish(a.dw_acc, p->nblocks, tapsrc, 27, d->Cout, d->Cin, d->Cout, d->Cin, sCo, sCi, dw, dbias_grad ? dbias_sums : nullptr,
                         dbias_grad, d->Cout, dbias_stride, stream);
}
