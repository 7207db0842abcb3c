/* stroke_amd.h -- C ABI of libstroke_amd.so (MI355X / gfx950 kernels).
 *
 * Drop-in boundary for the 3-D U-Net / CAE hot path of
 * multimodallearning/stroke-prediction.  The reference has no native code: each
 * entry point replaces the ATen/cuDNN kernels that the cited torch.nn call sites
 * reach implicitly.
 *
 * Conventions
 *  - plain C: raw device pointers, ints, floats; no C++ or torch types;
 *  - every call ENQUEUES on the given hipStream_t and returns; nothing here
 *    synchronises, allocates or frees device memory (caller owns all buffers);
 *  - return 0 on success, SP_EINVAL (bad shapes/pointers/alignment) or SP_EHIP
 *    (launch failure); text via sp_last_error() (thread-local);
 *  - activations are "channels-last-3d": [B][D][H][W][CP] with CP (channel
 *    pitch) a multiple of 8; dtype SP_BF16 (fast) or SP_F32 (parity mode, the
 *    convolutions then run split-bf16 x3 MFMA, ~fp32 accurate);
 *  - statistics / parameter buffers are fp32 unless stated; reduction
 *    accumulators are fp64 ("sums" arguments) and must be zeroed by the caller.
 *    The elementwise kernels (sp_bn_stats, sp_bn_bwd_reduce, sp_bn_act_bwd, sp_maxpool2_fwd, sp_upsample2_*,
 *    sp_crop_copy, sp_pool_skip_act_bwd, sp_out_grad_to_cl, sp_first_wgrad_fused) add into SP_REDUCE_ROWS replica
 *    rows of their accumulator, sums[SP_REDUCE_ROWS][CP (x2 for statistics)] with CP the channel pitch of the tensor
 *    reduced over (same-line fp64 atomics from different workgroups serialise at ~20 ns each on the 8-XCD part);
 *    the consumers (sp_bn_finalize / sp_bn_bwd_finalize with nrep >= SP_REDUCE_ROWS, sp_wgrad_finish* with
 *    dbias_stride = CP) add the rows.
 */
#ifndef STROKE_AMD_H
#define STROKE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sp_stream_t; /* hipStream_t */

enum { SP_OK = 0, SP_EINVAL = -1, SP_EHIP = -2 };
enum { SP_REDUCE_ROWS = 8 };
/* row pitch (doubles) of the Dice accumulator sums[SP_REDUCE_ROWS][SP_DICE_PITCH(C)]: whole 128-byte lines */
#define SP_DICE_PITCH(C) ((3 * (C) + 15) / 16 * 16)
enum { SP_BF16 = 0, SP_F32 = 1,
       /* bf16 PAIR: value = hi + lo with hi = bf16(value) and lo = bf16(value - hi), stored as TWO bf16 tensors of the same
        * shape (the hi tensor is what the bf16 kernels of the backward pass read; the lo tensor lies lo_delta bytes behind it).
        * ~17 significand bits.  Forward kernels of the "bf16x3" precision mode take and write pairs and multiply them with
        * hi/lo weight fragments on three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi, fp32 accumulate). */
       SP_HL = 2 };
enum { SP_ACT_NONE = 0, SP_ACT_LEAKY = 1, SP_ACT_ELU = 2, SP_ACT_SIGMOID = 3 };

int sp_version(void);
/* copies the calling thread's last error text (NUL-terminated) into buf */
void sp_last_error(char* buf, size_t n);

/* ------------------------------------------------------------------ implicit-GEMM convolution
 * One kernel serves nn.Conv3d forward (Unet3D.py:19,22; Cae3D.py:41-74,186-218), its data
 * gradient, and nn.ConvTranspose3d (Cae3D.py:178-204) -- the host plans taps, padding and the
 * K order (tables below) and the kernel is a table-driven LDS-tiled MFMA implicit GEMM:
 *   y[b,o,co] = act( bias[co] + sum_{tap,ci} w[co,ci,tap] * xin[b, o*s + o0 + tapoff, ci] )
 * with xin = x*in_scale[ci]+in_shift[ci] inside the input volume (BatchNorm applied on load,
 * Unet3D.py:18,21) and 0 outside it (zero padding AFTER the norm, Cae3D.py:41).
 */
/* BatchNorm finalize folded into the kernel that consumes its result (sp_conv_prep_folded_bn, sp_first_prep_bn): the arguments of
 * sp_bn_finalize.  Every workgroup of the consumer derives scale / shift of all C channels itself (same arithmetic, same order:
 * identical values everywhere); workgroup 0 also writes scale / shift / mean / invstd and updates the running statistics. */
typedef struct sp_bn_fin_args {
  const double* sums;      /* [nrep][CP][2] (sum x, sum x^2), replicas are added; unused when !training */
  const float* gamma;      /* [C]; NULL = no BatchNorm folded in (the plain entry point's behaviour) */
  const float* beta;
  float* running_mean;     /* [C], updated when training (momentum, unbiased variance) */
  float* running_var;
  float* scale;            /* [CP] outputs (what the backward reads): scale, shift, mean, invstd */
  float* shift;
  float* mean;
  float* invstd;
  double count;            /* voxels per channel */
  float momentum, eps;
  int32_t nrep, training, C, CP;
} sp_bn_fin_args;

/* BatchNorm BACKWARD finalize folded into its consumer (sp_conv3d_zm with stats_mode 2): the arguments of sp_bn_bwd_finalize.
 * dx = c0 g + c1 x + c2 with c0 = gamma invstd, c1 = -gamma invstd^2 dgamma / N, c2 = -c0 dbeta / N - c1 mean,
 * dgamma = (S2 - mean S1) invstd, dbeta = S1, (S1, S2) = (sum g, sum g x) = sums added over the replicas. */
typedef struct sp_bn_bwd_args {
  const double* sums;      /* [nrep][CP][2]; NULL = unused */
  const float* gamma;      /* [C] */
  const float* mean;       /* [CP] */
  const float* invstd;     /* [CP] */
  float* dgamma;           /* [C] += pscale * dgamma (workgroup 0), or NULL */
  float* dbeta;
  float* coef;             /* optional output [3][CP] (workgroup 0), or NULL */
  double count;
  float pscale;
  int32_t nrep, C, CP;
} sp_bn_bwd_args;

typedef struct sp_conv_args {
  /* tensors */
  const void* x;         /* [B][Di][Hi][Wi][CPi] */
  void* y;               /* [B][YD][YH][YW][CPo] */
  const void* wfrag_hi;  /* weight fragments from sp_conv_prep_weights */
  const void* wfrag_lo;  /* low halves (SP_F32 mode) or NULL */
  const float* in_scale; /* [CPi] or NULL (= no affine on load) */
  const float* in_shift; /* [CPi] */
  const float* bias;     /* [NTtot*16] padded with zeros, or NULL */
  double* stats;         /* [CPo][2] sum / sum-of-squares of the outputs, or NULL */
  const int32_t* ktab;   /* [steps_per_group*4] LDS byte offset of K-octet (step, lane group) */
  int32_t dtype_in, dtype_out;
  /* geometry */
  int32_t B, Di, Hi, Wi, CPi;
  int32_t Do, Ho, Wo;          /* logical output grid of this launch */
  int32_t YD, YH, YW, CPo;     /* full output tensor */
  int32_t osD, osH, osW;       /* output coordinate = o*os + oo (transposed-conv parity classes) */
  int32_t ooD, ooH, ooW;
  int32_t Cout;                /* real output channels; channels >= Cout are written as 0 */
  int32_t sD, sH, sW;          /* input step per output step */
  int32_t o0D, o0H, o0W;       /* input coordinate of (output 0, tap offset 0), may be negative */
  /* tiling (host plan) */
  int32_t TD, TH;              /* output tile = TD x TH rows of 16 voxels; TD*TH == 4*MT */
  int32_t ITD, ITH, ITW;       /* staged input tile extent (voxels) */
  int32_t MT, NT;              /* register blocking: M tiles per wave, N (cout/16) tiles per block */
  int32_t NTtot;               /* total cout tiles (grid.y = NTtot/NT) */
  int32_t ngroups;             /* channel groups staged one after the other */
  int32_t octs_per_group;      /* 8-channel octets per group */
  int32_t opp;                 /* octets per LDS plane */
  int32_t vsb;                 /* bytes per voxel inside a plane */
  int32_t plane_bytes;
  int32_t lo_offset;           /* byte offset of the low-half planes (SP_F32), else 0 */
  int32_t steps_per_group;     /* K steps (of 32) per group */
  int32_t lds_bytes;
  int32_t act;
  float act_param;
  int32_t dma;                 /* 1: LDS-DMA staging (bf16 in, in_scale NULL, lane-linear planes, +1 KiB LDS slack) */
  int32_t zfill;               /* dma only: taps can leave the input volume -> zero those chunks */
  int32_t persist;             /* dma only: 1/2 allow the persistent double-buffered variant where it applies;
                                  3: z-marching ring variant -- ktab then holds (in-plane byte offset | dz) per entry and
                                  ITH_zs the staged plane height (32 output rows + kernel extent - 1) */
  const void* aux;             /* stats_mode 1: tensor shaped like y (the layer input x of a data gradient) */
  int32_t stats_mode;          /* 0: stats = (sum y, sum y^2);  1: stats = (sum y, sum y*aux) -- BatchNorm backward sums
                                  fused into the dgrad epilogue (DMA kernel only) */
  int32_t stats_nrep;          /* power of two >= 1: stats is [stats_nrep][CPo][2]; workgroup b adds to replica b % nrep
                                  (tens of thousands of same-address fp64 atomics otherwise serialise at the memory side) */
  int32_t ITH_zs;              /* persist == 3 only */
  int64_t x_plane;             /* dma kernel: != 0 -> x is plane-major [CPi/16][B][D][H][W][16] with this many elements per
                                  plane (concat buffers written by sp_upsample2_crop_cat_fwd); 0 -> channels-last */
  /* ---- fp8 kernel (sp_conv3d_zm8) only; zero elsewhere */
  void* y8;                    /* optional second output: e4m3 plane-major copy of y, [Cout/16][B][YD][YH][YW][16 bytes] */
  int64_t y8_plane;            /* bytes per 16-channel plane of y8 (>= B*YD*YH*YW*16) */
  const float* f8_wscale;      /* [NT*16] epilogue multiplier per output channel (sp_conv_prep_f8: 2^-k of the weight row,
                                  times the reciprocal operand scale of a data gradient) */
  float y8_scale;              /* y8 = e4m3(y8_scale * y) */
  int32_t f8_bin;              /* B operand (x) format: 0 = e4m3 (forward), 1 = e5m2 (data gradient: x = quantised dz) */
  /* ---- batched passes (sp_conv3d_igemm / _multi only; 0 elsewhere) */
  int32_t group_batch;         /* > 0: samples [g*group_batch, (g+1)*group_batch) are BatchNorm group g: in_scale / in_shift of
                                  the group at + g*CPi, its statistics rows at stats + g*stats_nrep*CPo*2 */
  /* ---- fp8 kernel: several output-channel slices of an op in one launch (0 / 1: one slice) */
  int32_t nslices;             /* > 1: slice s computes output channels [s*Cout, (s+1)*Cout) -- y, bias, f8_wscale, stats are the
                                  pointers of slice 0 (the others follow at + s*Cout channels), y8 at + s*NT planes */
  int64_t slice_wfrag_stride;  /* bytes between the weight fragments (wfrag_hi) of consecutive slices */
  /* ---- bf16 pairs (dtype_in / dtype_out = SP_HL; sp_conv3d_zm and sp_conv3d_igemm forward kernels; 0 elsewhere) */
  int64_t x_lo_delta;          /* bytes from x (hi halves) to the lo halves (a tensor of the same shape and layout) */
  int64_t y_lo_delta;          /* the same for y */
  /* ---- BatchNorm folded per group into a PADDED convolution (sp_conv3d_zm forward with the ELU epilogue; 0 / NULL elsewhere):
   * zero padding applies after the normalisation x^ = s x + t, so next to the folded weights W s the bias depends on which taps
   * of an output voxel fall into the padding: bias_tab[g][class][CPo] = b + sum over the VALID taps of W t, class = (cz * ny + cy)
   * * nx + cx, c = o for o < pad, pad inside, pad + 1 + (o - (n_out - pad)) behind: 2 pad + 1 classes per axis
   * (sp_conv_prep_folded_groups writes fragments and table) */
  const float* bias_tab;       /* NULL: the plain bias */
  int32_t bias_tab_gstride;    /* floats between the tables of consecutive groups */
  int32_t pad_;
  int64_t wfrag_gstride;       /* bytes between the weight fragments (wfrag_hi) of consecutive groups (0: one set) */
  /* ---- sp_conv3d_zm, stats_mode 2: the data gradient g = conv^T(dz_above, W) of the SECOND convolution of a block is not stored;
   * its epilogue applies the BatchNorm backward of that convolution's input BatchNorm and the activation derivative of the block's
   * first convolution, dz = (c0 g + c1 x + c2) act'(x) with x = aux (the first convolution's output = the BatchNorm's input), and
   * writes dz to y -- what sp_bn_act_bwd would have made of a stored g in a second pass over both tensors (Unet3D.py:18-24 backward).
   * The coefficients come from bnb (the kernel finalizes them itself: the weight gradient's finish kernel has left the sums);
   * dz_sums[SP_REDUCE_ROWS][CPo] += sum over voxels of dz (the first convolution's bias gradient / folded weight-gradient term). */
  sp_bn_bwd_args bnb;
  double* dz_sums;
  /* ---- sp_conv3d_zm forward layers (P, NT) = (1, 1) / (2, 2): MaxPool3d(2, 2) (floor) of the output rides in the epilogue
   * (Unet3D.py:59,62): pool_y [B][YD/2][YH/2][YW/2][CPo] (SP_HL: the lo half pool_lo_delta bytes behind) is written next to y, and
   * `stats` then receives the statistics of the POOLED tensor -- the next block's BatchNorm input -- instead of y's */
  void* pool_y;
  int64_t pool_lo_delta;
  /* ---- sp_conv3d_zm data gradients of a layer whose input is a channel concatenation (Unet3D.py:66-67,71-72): output tiles
   * [0, split_nt) go to y (channel pitch CPo), the others to y2 [B][YD][YH][YW][CPo2] -- two dense tensors for the two consumers of
   * the gradient (upsample backward / pool + skip backward) from one launch */
  void* y2;
  int32_t split_nt, CPo2;
  /* ---- sp_conv3d_zm "plane-serial" forward layers (round 5; 48 -> 16 in the pair mode, 96 -> 32): pser_planes = CPi / 16 > 0 --
   * the march takes one 16-channel plane per sub-step and streams that plane's weight fragments through LDS; wfrag_hi / _lo in
   * the order [plane][(dz KS1 + s) NT + n] with KS1 = 5 (the one-plane K table), ktab = the one-plane table
   * (runtime/plan.py:zm_pser_plan); MT / slots / waves: sp_conv3d_zm_config_ps */
  int32_t pser_planes, pad2_;
} sp_conv_args;

int sp_conv3d_igemm(const sp_conv_args* a, sp_stream_t stream);
/* Weight fragments and bias tables of a stride-1 3x3x3 convolution with padding (padD, padH, padW) <= 2 behind a BatchNorm whose
 * scale / shift differ per group of the batch (the CAE's batched passes): group g's fragments (W s_g, bf16, K order of kmap) at
 * wfrag + g * frag_gstride bytes and its table (see sp_conv_args.bias_tab) at bias_tab + g * ncls * CoutPad floats; the rows
 * (scale, -, shift) of group g at coef + g * coef_gstride with pitch coef_pitch.  One launch.  Cae3D.py:41-70, 186-218. */
int sp_conv_prep_folded_groups(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap, int32_t nsteps,
                               int32_t NTtot, void* wfrag, int64_t frag_gstride, const float* coef, int32_t coef_gstride,
                               int32_t coef_pitch, int32_t G, const float* bias, int32_t padD, int32_t padH, int32_t padW,
                               float* bias_tab, int32_t CoutPad, sp_stream_t stream);
/* n (<= 8) sub-convolutions of one op -- the parity classes of a stride-2 transposed convolution (Cae3D.py:178-204) or of a
 * strided convolution's data gradient -- in ONE launch when they share register blocking, data types and the register-staged
 * kernel (dma = 0); otherwise the classes are launched one after the other.  Same result either way. */
int sp_conv3d_igemm_multi(const sp_conv_args* args, int32_t n, sp_stream_t stream);
/* The same n classes in ONE pass over the output (csrc/sp_conv_par.hip; bf16 channels-last in and out): a workgroup owns a tile
 * of the class grid and walks all classes for it, so the s x s x s output voxels of a class-grid voxel leave together (whole
 * lines) and the input is fetched once; operands are gathered straight from global memory / L2, no LDS staging.
 * args[i]: what sp_conv3d_igemm takes for class i (its wfrag_hi in the K order of its kmap, ngroups * steps_per_group K steps,
 * Do / Ho / Wo, oo*, o0*; tensors, strides, NTtot, bias, act, stats / stats_mode / aux / group_batch of class 0 apply to all).
 * gtab (device): two int32 per K slot of every class, class after class -- byte offset of the slot's (tap, octet) from the
 * lane's base input voxel, and oz | (4 + oy) << 8 | (8 + ox) << 16 with the tap's offsets (0..2) from the class origin o0;
 * gofs[i] (host, n + 1 entries): first slot of class i.  zeros: >= 16 readable zero bytes (out-of-volume taps).
 * Replaces nn.ConvTranspose3d forward (Cae3D.py:178-204) and the data gradient of the strided nn.Conv3d layers (Cae3D.py:45-64). */
int sp_conv3d_par(const sp_conv_args* args, int32_t n, const void* gtab, const int32_t* gofs, const void* zeros, sp_stream_t stream);

/* The same operation (nn.Conv3d(3, stride 1, padding 0) forward, Unet3D.py:19,22, or its data gradient) on the
 * output-stationary z-marching kernel (csrc/sp_conv_zm.hip): a workgroup marches through the INPUT planes of a column of
 * NW*MT x 16 output voxels, each plane staged once by LDS-DMA and fed to the three output planes it contributes to.
 * Uses of sp_conv_args: x, y, bias, stats / stats_nrep (stats_mode 0), geometry, o0*, act (LEAKY / NONE), x_plane, Cout,
 * CPi = 16 P, NT = NTtot = Cout / 16, MT (rows per wave) as sp_conv3d_zm_config(P, NT) reports; ktab and wfrag_hi in the
 * kernel's own K order (runtime/plan.py:zm_plan): ktab[s*4+g] = byte offset of the octet inside a ring slot, fragments
 * [(dz*KS + s)*NT + n] from sp_conv_prep_weights / sp_conv_prep_folded with the matching kmap.  in_scale must be NULL (the
 * BatchNorm is folded into the weights).  zeros: >= 16 readable zero bytes on the device (source of padding chunks). */
int sp_conv3d_zm(const sp_conv_args* a, const void* zeros, sp_stream_t stream);
/* (input planes P = Cin/16, output tiles NT = Cout/16) -> rows per wave, ring slots and waves per workgroup of the kernel that
 * exists for the pair (a workgroup covers NW*MT x 16 output voxels per plane); returns SP_EINVAL when there is none */
int sp_conv3d_zm_config(int32_t P, int32_t NT, int32_t* MT, int32_t* NSLOT, int32_t* NW);
/* the plane-serial instances (sp_conv_args.pser_planes): NT output tiles, hl = bf16 pairs */
int sp_conv3d_zm_config_ps(int32_t NT, int32_t hl, int32_t* MT, int32_t* NSLOT, int32_t* NW);

/* ------------------------------------------------------------------ bf16 pairs (SP_HL): the forward pass of the "bf16x3" mode
 * north_star asks for logits within 1e-3 of the CPU reference; bf16 storage (8 significand bits per activation) cannot give
 * that and fp32 storage with split MFMAs costs 4.7x the bf16 step.  In this mode every activation of the FORWARD pass is a pair
 * of bf16 tensors (hi = bf16(v), lo = bf16(v - hi): ~17 bits), every forward convolution multiplies pairs with hi/lo weight
 * fragments on three bf16 MFMAs per product (sp_conv3d_zm / sp_conv3d_igemm with dtype_in = dtype_out = SP_HL, x_lo_delta,
 * y_lo_delta, wfrag_lo), and the BACKWARD pass is the bf16 one, unchanged, on the hi tensors (which are exactly the tensors the
 * bf16 mode would have stored).  Replaces Block3x3x3.forward / Unet3D.forward (Unet3D.py:14-27,56-79) at fp32-like accuracy.
 * The entry points below are the pair forms of the forward-only kernels; *_lo_delta = byte distance from a hi tensor to its lo
 * tensor (same shape and layout, 16-byte aligned). */
int sp_conv3d_zm_config_hl(int32_t P, int32_t NT, int32_t* MT, int32_t* NSLOT, int32_t* NW);
/* first BatchNorm's statistics of the fp32 NCDHW input as it is (sp_bn_stats_ncdhw rounds to the 16-bit type first) */
int sp_bn_stats_ncdhw_f32(const float* x, int32_t B, int32_t C, int64_t DHW, int32_t CP, double* sums, int32_t nrep,
                          sp_stream_t stream);
/* first layer (BatchNorm3d(2) -> Conv3d(2, 16 | 32, 3) -> act, Unet3D.py:18-20): hi + lo weight fragments, y / y_lo pair */
int sp_first_prep_hl(const float* w, const float* b, const float* scale, const float* shift, void* wfrag_hi, void* wfrag_lo,
                     float* bias_f, int32_t Cout, sp_stream_t stream);
int sp_first_conv_fwd_hl(const float* x, int32_t B, int32_t D, int32_t H, int32_t W, const void* wfrag_hi, const void* wfrag_lo,
                         const float* bias_f, int32_t act, float act_param, void* y, void* y_lo, double* stats, int32_t nrep,
                         int32_t Cout, sp_stream_t stream);
/* MaxPool3d(2,2) (Unet3D.py:39,41) of the pair values, written as a pair; stats as sp_maxpool2_fwd */
int sp_maxpool2_fwd_hl(const void* x, int64_t x_lo_delta, void* y, int64_t y_lo_delta, int32_t B, int32_t D, int32_t H, int32_t W,
                       int32_t CP, double* stats, sp_stream_t stream);
/* Upsample x2 + crop + concat (Unet3D.py:64-72) on pairs; cat_plane as for sp_upsample2_crop_cat_fwd (elements of one half) */
int sp_upsample2_crop_cat_fwd_hl(const void* low, int64_t low_lo_delta, int32_t CPu, const void* skip, int64_t skip_lo_delta, int32_t CPs,
                                 void* cat, int64_t cat_lo_delta, int32_t CPd, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds,
                                 int32_t Hs, int32_t Ws, int64_t cat_plane, double* stats, sp_stream_t stream);
/* classify head (Unet3D.py:49-54,75-77) on a pair input: fp32 arithmetic, seg as sp_head_fwd */
int sp_head_fwd_hl(const void* x, int64_t x_lo_delta, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                   const float* b1, int32_t CH, const float* w2, const float* b2, int32_t NC, float slope, float* seg,
                   sp_stream_t stream);

/* ------------------------------------------------------------------ fp8 (BASELINE.json configs[4]: "4-scale U-Net ... fp8 MFMA")
 * The same z-marching operation on v_mfma_f32_16x16x128_f8f6f4 (csrc/sp_conv_zm8.hip): x is an fp8 PLANE-MAJOR tensor
 * [CPi/16][B][Di][Hi][Wi][16 bytes] (x_plane = bytes per plane), e4m3 for a forward convolution, e5m2 (f8_bin = 1) for a data
 * gradient; weights are e4m3 fragments from sp_conv_prep_f8 with one power-of-two scale per output channel, undone by the
 * epilogue (f8_wscale); accumulation, bias, LeakyReLU and the BatchNorm statistics are fp32; y is bf16 channels-last as for
 * sp_conv3d_zm, y8 (optional) its e4m3 plane-major copy for the next fp8 convolution.  ktab[(s*4+g)*2+h] = byte offset of
 * chunk h of lane group g in K step s inside a ring slot (runtime/plan.py:zm8_plan).  Replaces nn.Conv3d(3, padding 0)
 * (Unet3D.py:19,22 inside the 4-scale topology Unet3D.py:95-146) for 32 <= Cin <= 96. */
int sp_conv3d_zm8(const sp_conv_args* a, const void* zeros, sp_stream_t stream);
/* Input-channel groups: a convolution with more input planes than an fp8 instance holds runs one sp_conv3d_zm8 per group of
 * planes with dtype_out = SP_F32 (y = the group's fp32 partial sums [B*YD*YH*YW][CPo], no bias / activation / statistics), then
 * y[m][c] (bf16) = act(sum_g partial[g][m][c] + sum_g bias[g*bias_stride + c]); stats[rep][CP][2] += (sum y, sum y^2). */
int sp_conv_partial_finish(const float* partial, int32_t ngroups, int64_t nvox, int32_t CP, const float* bias /* or NULL */,
                           int32_t bias_stride, int32_t act, float act_param, void* y, double* stats /* or NULL */,
                           int32_t stats_nrep, void* y8 /* or NULL: e4m3 plane-major copy of y, [CP/16][nvox][16 bytes] */,
                           int64_t y8_plane, sp_stream_t stream);
int sp_conv3d_zm8_config(int32_t P, int32_t NT, int32_t* MT, int32_t* NSLOT, int32_t* NW);
/* fp32 weights -> e4m3 A fragments in the plan's K order.  kmap[(step*4+g)*2+h] = (src_tap << 16) | input 16-channel plane, or
 * -1 (zero chunk); element (co, ci, tap) = w[co*sCo + ci*sCi + tap] * fold_scale[ci] (fold_scale may be NULL).  Per output
 * channel: 2^k = largest power of two with |row| * 2^k <= 224; winv[co] = out_scale / 2^k; bias_out[co] (may be NULL) =
 * bias[co] + sum_{ci,tap} w * fold_shift[ci] in fp32 (BatchNorm folded into an un-padded convolution, exact). */
int sp_conv_prep_f8(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap,
                    int32_t nsteps, int32_t NT, void* wfrag, const float* fold_scale, const float* fold_shift,
                    int32_t ntaps, const float* bias, float* bias_out, float* winv, float out_scale, sp_stream_t stream);
/* the same for n weight tensors / output-channel slices in ONE launch (a sliced op packs one item per slice; the data-gradient
 * weights of a whole network depend on the parameters only).  items_dev: DEVICE array; max_rows = max over the items of NT*16. */
typedef struct sp_f8_prep_item {
  const float* w;
  int64_t sCo, sCi;
  int32_t Cout, Cin;
  const int32_t* kmap;
  int32_t nsteps, NT;
  void* wfrag;
  const float* fold_scale;   /* or NULL */
  const float* fold_shift;   /* or NULL */
  const float* bias;         /* or NULL */
  float* bias_out;           /* or NULL */
  float* winv;
  int32_t ntaps;
  float out_scale;
} sp_f8_prep_item;
int sp_conv_prep_f8_batch(const sp_f8_prep_item* items_dev, int32_t n, int32_t max_rows, sp_stream_t stream);
/* dst = fp8(scale * src): src bf16 channels-last [nvox][CP] (src_plane = 0) or plane-major [CP/16][..][16] (src_plane =
 * elements per plane); dst plane-major [CP/16][nvox][16 bytes] with dst_plane bytes per plane; fmt 0 = e4m3 (saturating at
 * 448), 1 = e5m2 (57344).  The operand of sp_conv3d_zm8 where no producer wrote it (pooled / concatenated tensors, dz). */
int sp_quantize_f8(const void* src, int32_t CP, int64_t src_plane, void* dst, int64_t dst_plane, int64_t nvox, int32_t fmt,
                   float scale, sp_stream_t stream);

/* ------------------------------------------------------------------ self-sufficient entry points for the hot convolution
 * nn.Conv3d(Cin, Cout, 3, stride 1, padding 0) of Block3x3x3 (Unet3D.py:19,22) and its data gradient on the z-marching
 * kernel, for callers that have this header and nothing else (no runtime/plan.py): describe the layer, ask for the plan
 * (workspace size, output extents), allocate `workspace_bytes` of device memory, init once, set weights whenever they
 * change, run.  Tensors are bf16 channels-last [B][D][H][W][C] with C a multiple of 16 (zero-padded channels).
 *   grad = 0: x = [B][D][H][W][Cin]          -> y = [B][D-2][H-2][W-2][Cout]   y = act(conv(x, w) + bias)
 *   grad = 1: x = dz [B][D-2][H-2][W-2][Cout] -> y = [B][D][H][W][Cin]          y = conv_transpose(dz, w)
 * (D, H, W are always the extents of the convolution's INPUT).  Pairs (Cin/16, Cout/16) of the op without a kernel
 * (sp_conv3d_zm_config) are refused by sp_conv3d_plan: those layers need sp_conv3d_igemm and a host-built tile plan. */
typedef struct sp_conv3d_desc {
  int32_t B, Cin, Cout, D, H, W;
  int32_t grad;
  /* round 5 (zero = the un-padded nn.Conv3d above): the CAE's layers (Cae3D.py:41-70, 178-218) on the same kernel --
   * padD / padH / padW in 0..2: nn.Conv3d(Cin, Cout, 3, stride 1, padding (padD, padH, padW)); zero padding comes from the
   *   kernel's zero page, so a BatchNorm in FRONT of a padded layer cannot be folded by sp_conv3d_set_weights (the padding is
   *   applied after the normalisation: give the kernel the normalised tensor, or sp_conv_prep_folded_groups' bias table);
   * transposed = 1: nn.ConvTranspose3d(Cin, Cout, 3, stride 1, padding pad) -- w is [Cin][Cout][3][3][3], the output extent
   *   D + 2 - 2 padD (the "full" correlation with the mirrored kernel, cropped by the padding); grad must be 0;
   * act of sp_conv3d_run may be SP_ACT_ELU (act_param = alpha) for (Cin/16, Cout/16) pairs of up to 2 input planes. */
  int32_t padD, padH, padW;
  int32_t transposed;
} sp_conv3d_desc;
typedef struct sp_conv3d_plan_t {
  int32_t cin_op, cout_op;             /* channels the op reads / writes (swapped for the data gradient) */
  int32_t P, NT, MT, NW, NSLOT, KS, nsteps, ITH;
  int32_t Di, Hi, Wi, Do, Ho, Wo, o0;  /* extents of x and y, input coordinate of (output 0, tap 0) (o0: the D axis') */
  int32_t o0H, o0W, mirror;            /* ... of the H and W axes; mirror: the kernel's taps are the weight's, point-mirrored */
  int64_t x_elems, y_elems;            /* bf16 elements of x and y */
  int64_t workspace_bytes;
  int64_t off_zero, off_ktab, off_kmap, off_bias, off_wfrag;   /* layout of the workspace */
} sp_conv3d_plan_t;
int sp_conv3d_plan(const sp_conv3d_desc* d, sp_conv3d_plan_t* plan);                 /* host only */
/* host only: the kernel's K tables, ktab[4*KS] and kmap[12*KS] (what sp_conv3d_init uploads; runtime/plan.py:zm_plan) */
int sp_conv3d_tables(const sp_conv3d_desc* d, const sp_conv3d_plan_t* plan, int32_t* ktab, int32_t* kmap);
int sp_conv3d_init(const sp_conv3d_desc* d, const sp_conv3d_plan_t* plan, void* workspace, sp_stream_t stream);
/* w: fp32 [Cout][Cin][3][3][3] on the device (nn.Conv3d layout); bias [Cout] or NULL; bn_scale / bn_shift [Cin] or NULL:
 * a BatchNorm in front of the convolution folded into weights and bias (exact: no padding).  grad = 1: w only. */
int sp_conv3d_set_weights(const sp_conv3d_desc* d, const sp_conv3d_plan_t* plan, void* workspace, const float* w,
                          const float* bias, const float* bn_scale, const float* bn_shift, sp_stream_t stream);
/* act: SP_ACT_NONE or SP_ACT_LEAKY; stats: NULL or [stats_nrep][Cout][2] fp64 (sum, sum of squares) accumulators;
 * x_plane: 0 for channels-last x, else elements per 16-channel plane of a plane-major x */
int sp_conv3d_run(const sp_conv3d_desc* d, const sp_conv3d_plan_t* plan, const void* workspace, const void* x, void* y,
                  int32_t with_bias, int32_t act, float act_param, double* stats, int32_t stats_nrep, int64_t x_plane,
                  sp_stream_t stream);

/* The layer's WEIGHT gradient for the same header-only caller (grad is ignored in `d`): dw[co][ci][27] (fp32, nn.Conv3d layout,
 * ACCUMULATED into) += sum over voxels of dz[v][co] * xin[v + tap][ci] with x the layer's bf16 input [B][D][H][W][Cin] and dz the
 * bf16 output gradient [B][D-2][H-2][W-2][Cout].  bn_scale / bn_shift (the BatchNorm in front of the conv, as sp_bn_finalize
 * writes them): xin = scale * x + shift is folded into the finish pass (dW = scale * acc + shift * sum dz) and needs dbias_sums
 * = sum over voxels of dz per output channel (fp64: one row, dbias_stride = 0, or SP_REDUCE_ROWS replica rows of dbias_stride
 * doubles as sp_bn_act_bwd & co fill them); dbias_grad (or NULL) += sum dz; w_for_bn + bn_sums (or NULL): the BatchNorm-backward
 * sums (sum g, sum g*x) of the layer's input gradient g read off the accumulator (bn_nrep replica rows of [Cin][2] doubles,
 * zeroed by the caller), which makes the data gradient of a first layer unnecessary.  x_plane as for sp_conv3d_run. */
typedef struct sp_conv3d_wgrad_plan_t {
  int32_t CoT, CiT, nblocks, Do, Ho, Wo;
  int64_t workspace_bytes, off_taps, off_tapsrc, off_acc;
} sp_conv3d_wgrad_plan_t;
int sp_conv3d_wgrad_plan(const sp_conv3d_desc* d, sp_conv3d_wgrad_plan_t* p);
int sp_conv3d_wgrad_init(const sp_conv3d_desc* d, const sp_conv3d_wgrad_plan_t* p, void* workspace, sp_stream_t stream);
int sp_conv3d_wgrad_run(const sp_conv3d_desc* d, const sp_conv3d_wgrad_plan_t* p, void* workspace, const void* x, const void* dz,
                        float* dw, const float* bn_scale /* or NULL */, const float* bn_shift, const double* dbias_sums /* or NULL */,
                        int32_t dbias_stride, float* dbias_grad /* or NULL */, const float* w_for_bn /* or NULL */,
                        double* bn_sums /* or NULL */, int32_t bn_nrep, int64_t x_plane, sp_stream_t stream);

/* ------------------------------------------------------------------ FC-like layers: split-K convolution without LDS
 * The 800 <-> 100 channel layers around the CAE's latent (Cae3D.py:72-76, 178-180): a few hundred output voxels per sample,
 * K = taps x Cin in the tens of thousands.  One workgroup per (64 output voxels, 8 output tiles, TAP); a second kernel sums the
 * per-tap fp32 partials and applies bias / activation / statistics.  Single dense correlation only (out_stride 1):
 *   y[b,o,co] = act(bias[co] + sum_{tap,ci} w[co,ci,src(tap)] * xin[b, o*s + o0 + taps[tap], ci])
 * wfrag: sp_conv_prep_weights with kmap[(tap*spt + q)*4 + g] = (src_tap << 16) | (4q + g)  (spt = ceil(CPi/32) rounded up to a multiple of 4; -1 past the
 * last octet), NTtot = ceil(Cout/16).  partial: sp_conv_fc_workspace floats.  stats as in sp_conv_args (stats_mode 1: aux is
 * a bf16 tensor shaped like y). */
typedef struct sp_conv_fc_args {
  const void* x;           /* bf16 [B][Di][Hi][Wi][CPi] */
  void* y;                 /* [B][Do][Ho][Wo][CPo], dtype_out */
  const void* wfrag;
  const float* in_scale;   /* [CPi] BatchNorm on load (inside the volume only), or NULL */
  const float* in_shift;
  const float* bias;       /* [>= Cout] or NULL */
  double* stats;           /* [stats_nrep][CPo][2] or NULL */
  const void* aux;
  float* partial;
  const int32_t* taps;     /* [ntap][3] */
  int32_t B, Di, Hi, Wi, CPi, Do, Ho, Wo, CPo, Cout;
  int32_t sD, sH, sW, o0D, o0H, o0W;
  int32_t ntap;
  int32_t act;
  float act_param;
  int32_t stats_mode, stats_nrep, dtype_out;
  int64_t x_plane;         /* != 0: x is plane-major [CPi/16][B][D][H][W][16] with this many elements per plane */
  /* batched passes (the CAE): samples [g*group_batch, (g+1)*group_batch) are BatchNorm group g -- its statistics rows at
   * stats + g*stats_nrep*CPo*2, its in_scale / in_shift rows at + g*coef_gstride floats; one launch for all groups (0: one group) */
  int32_t group_batch;
  int32_t coef_gstride;
} sp_conv_fc_args;
int sp_conv_fc_workspace(int32_t B, int32_t Do, int32_t Ho, int32_t Wo, int32_t Cout, int32_t ntap, int64_t* floats);
int sp_conv_fc(const sp_conv_fc_args* a, sp_stream_t stream);

/* Re-pack fp32 weights into MFMA A-fragments in the plan's K order.
 * kmap[step*4+g] = (src_tap_index << 16) | cin_octet, or -1 for a padding octet.
 * element (co, ci, tap) is read from w[co*sCo + ci*sCi + tap]. */
int sp_conv_prep_weights(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin,
                         const int32_t* kmap, int32_t nsteps, int32_t NTtot,
                         void* wfrag_hi, void* wfrag_lo /* or NULL */,
                         const float* fold_scale /* [Cin] or NULL: w[co,ci,tap] *= fold_scale[ci] */,
                         sp_stream_t stream);
/* The same for `n` weight tensors in one launch (the data-gradient re-packs of a whole network depend on the parameters
 * only: one dispatch per step instead of one per layer).  items_dev: DEVICE array; max_blocks = max over the items of
 * ceil(nsteps*NTtot*64 / 256). */
typedef struct sp_prep_item {
  const float* w;
  int64_t sCo, sCi;
  int32_t Cout, Cin;
  const int32_t* kmap;
  int32_t nsteps, NTtot;
  void* wfrag_hi;
  void* wfrag_lo;            /* or NULL */
  const float* fold_scale;   /* or NULL */
  const float* bias;         /* or NULL: bias_out (when given) is then zeroed */
  float* bias_out;           /* or NULL; bias_pad floats: bias[0..bias_n) followed by zeros (what sp_conv_args.bias reads) */
  int32_t bias_n, bias_pad;
} sp_prep_item;
int sp_conv_prep_weights_batch(const sp_prep_item* items_dev, int32_t n, int32_t max_blocks, sp_stream_t stream);
/* BatchNorm folded into an un-padded convolution: bias_out[co] = bias[co] + sum_{ci,tap} w[co,ci,tap]*shift[ci]
 * (bias may be NULL = 0; bias_out has room for CoutPad entries, the tail is zeroed) */
int sp_conv_fold_bias(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, int32_t ntaps,
                      const float* bias, const float* shift, float* bias_out, int32_t CoutPad, sp_stream_t stream);
/* sp_conv_prep_weights (with fold_scale) and sp_conv_fold_bias in one launch */
int sp_conv_prep_folded(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap,
                        int32_t nsteps, int32_t NTtot, void* wfrag_hi, void* wfrag_lo, const float* fold_scale,
                        int32_t ntaps, const float* bias, const float* fold_shift, float* bias_out, int32_t CoutPad,
                        sp_stream_t stream);

/* sp_conv_prep_folded with the BatchNorm finalize (sp_bn_finalize) inside: fold_scale / fold_shift are computed from bn->sums by
 * every workgroup of the re-pack kernel and written to bn->scale / bn->shift by the first -- one launch per layer and step instead
 * of two on the forward's dependent chain (conv N -> statistics -> [finalize -> re-pack] -> conv N+1) */
int sp_conv_prep_folded_bn(const float* w, int64_t sCo, int64_t sCi, int32_t Cout, int32_t Cin, const int32_t* kmap,
                           int32_t nsteps, int32_t NTtot, void* wfrag_hi, void* wfrag_lo, int32_t ntaps, const float* bias,
                           float* bias_out, int32_t CoutPad, const sp_bn_fin_args* bn, sp_stream_t stream);
/* the same for the packed first layer (sp_first_prep_n / sp_first_prep_hl: wfrag_lo NULL or the lo fragments) */
int sp_first_prep_bn(const float* w, const float* b, void* wfrag, void* wfrag_lo, float* bias_f, int32_t Cout,
                     const sp_bn_fin_args* bn, sp_stream_t stream);

/* ------------------------------------------------------------------ weight gradient
 * dw[co,ci,tap] += sum_{b,o} dz[b,o,co] * xin[b, o*s + o0 + tapoff(tap), ci]
 * dw_acc layout: [ntap][CoutP16][CinP16] fp32.  parts == 0: one such block, zeroed by the caller, accumulated with
 * fp32 atomics.  parts == 1: `nblocks` such blocks, one WRITTEN (not accumulated) by each of the nblocks persistent
 * workgroups -- no atomics, deterministic; cross-XCD atomics cost 40-90 us per layer on MI355X.  sp_wgrad_finish
 * adds the blocks up and scatters them (accumulating) into the (Cout,Cin,k,k,k)-layout gradient buffer. */
typedef struct sp_wgrad_args {
  const void* x;         /* [B][Di][Hi][Wi][CPi] conv input (pre-norm) */
  const void* dz;        /* [B][Do][Ho][Wo][CPo] gradient at the conv output (pre-activation) */
  const float* in_scale; /* BatchNorm-on-load of x, or NULL */
  const float* in_shift;
  const float* dz_scale; /* affine on load of dz (transposed-conv roles: dz operand = normalised input), or NULL */
  const float* dz_shift;
  float* dw_acc;
  const int32_t* taps;   /* [ntap][3] input offsets (dz,dy,dx) added to o*s + o0 */
  int32_t dtype;
  int32_t B, Di, Hi, Wi, CPi, Do, Ho, Wo, CPo;
  int32_t sD, sH, sW, o0D, o0H, o0W;
  int32_t ntap;
  int32_t kD, kH, kW;    /* tap offsets satisfy 0 <= offset < k (sizes the staged halo) */
  int32_t CoT, CiT;      /* cout / cin tiles of 16 */
  int32_t nblocks;       /* persistent grid size */
  int32_t dma;           /* 1: bf16 LDS-DMA double-buffered path (stride 1, padding 0, 3x3x3, no affine on load) */
  int32_t tile_rows;     /* dma: 0 = choose, else force TZ*TY rows of 32 voxels per tile (tuning knob) */
  int32_t parts;         /* 1: dw_acc holds nblocks partial blocks (see above) */
  int32_t cib;           /* 0 = choose, else cin tiles (of 16) per workgroup: fewer planes per tile leave room for a
                            larger spatial tile (less halo re-read) at the price of re-reading dz per cin group */
  int64_t x_plane;       /* dma kernel, cib == 1: != 0 -> x is plane-major [CPi/16][B][D][H][W][16], elements per plane */
  int32_t zs;            /* dma kernel: 1 = allow the z-marching ring variant (cib == 1, parts == 1, Cout <= 32 per group) */
  int32_t groups;        /* pointwise kernel, parts == 1: > 1 -> the batch holds this many equal BatchNorm groups and partial block i covers
                            voxels of group i * groups / nblocks only (nblocks % groups == 0): sp_wgrad_finish_folded_groups.  (The
                            row-sliding 3x3x3 kernel is group-pure whenever nblocks % groups == 0.) */
} sp_wgrad_args;
int sp_conv3d_wgrad(const sp_wgrad_args* a, sp_stream_t stream);
/* BatchNorm folded out of the operand load (un-padded convolutions):
 * dw[co,ci,tap] += scale[ci]*dw_acc[tap][co][ci] + shift[ci]*dbias_sums[co] */
int sp_wgrad_finish_folded(float* dw_acc, int32_t nparts, const int32_t* tapsrc, int32_t ntap, int32_t CoP, int32_t CiP,
                           int32_t Cout, int32_t Cin, int64_t sCo, int64_t sCi, const float* scale,
                           const float* shift, const double* dbias_sums, float* dw, float* dbias_grad /* or NULL */,
                           const float* w_for_bn /* or NULL */, double* bn_sums /* or NULL */, int32_t bn_nrep,
                           int32_t bn_cp /* channel pitch of a bn_sums replica; <= 0: CiP */,
                           int32_t dbias_stride /* row pitch of the SP_REDUCE_ROWS replica rows of dbias_sums; <= 0: one row */,
                           sp_stream_t stream);
/* the same with dw_acc multiplied by acc_scale first (sp_conv3d_wgrad_f8: the accumulators carry the scale of the quantised dz) */
int sp_wgrad_finish_folded_scaled(float* dw_acc, int32_t nparts, const int32_t* tapsrc, int32_t ntap, int32_t CoP, int32_t CiP,
                                  int32_t Cout, int32_t Cin, int64_t sCo, int64_t sCi, const float* scale, const float* shift,
                                  const double* dbias_sums, float* dw, float* dbias_grad, const float* w_for_bn, double* bn_sums,
                                  int32_t bn_nrep, int32_t bn_cp, int32_t dbias_stride, float acc_scale, sp_stream_t stream);
/* fp8 weight gradient of nn.Conv3d(3, stride 1, padding 0) (Unet3D.py:19,22; csrc/sp_wgrad_f8.hip): x = the conv's input as
 * a plane-major e4m3 tensor [CiT][B][Di][Hi][Wi][16 bytes], dz = the output gradient as a plane-major e5m2 tensor
 * [CoT][B][Do][Ho][Wo][16 bytes] (quantised with a power-of-two scale S); CoT, CiT even.  Each of the nblocks persistent
 * workgroups WRITES one block [27][CoT*16][CiT*16] fp32 of dw_acc = S * partial sum; sp_wgrad_finish_folded_scaled(acc_scale =
 * 1/S) adds them up (and yields the BatchNorm-backward sums as for the bf16 kernels). */
typedef struct sp_wgrad_f8_args {
  const void* x;
  const void* dz;
  float* dw_acc;
  int32_t B, Di, Hi, Wi, Do, Ho, Wo;
  int32_t CoT, CiT;          /* cout / cin tiles of 16 */
  int32_t nblocks;           /* grid.x: persistent workgroups per 32 x 32 channel block */
  int64_t x_plane, dz_plane; /* bytes per 16-channel plane */
} sp_wgrad_f8_args;
int sp_conv3d_wgrad_f8(const sp_wgrad_f8_args* a, sp_stream_t stream);
/* bn_sums != NULL (first layer of a network: no input gradient wanted, so no data-gradient convolution is run):
 * also accumulate the BatchNorm-backward sums of the layer's input,  bn_sums[rep][ci][0] += sum_v g  and
 * bn_sums[rep][ci][1] += sum_v g*x  with g = conv_transpose(dz, w_for_bn), computed from the weight-gradient
 * accumulator:  sum_v g*x = sum_{co,tap} w*dw_acc,  sum_v g = sum_{co,tap} w*dbias_sums[co]  (un-padded conv). */
/* dw[co*sCo + ci*sCi + tapsrc[t]] += sum over the nparts blocks of dw_acc[t][co][ci].  With nparts == 1 (atomics
 * mode) both finish kernels also zero dw_acc for the next step.  Both, when dbias_grad != NULL, add the bias gradient dbias_grad[co] += dbias_sums[co]. */
int sp_wgrad_finish(float* dw_acc, int32_t nparts, const int32_t* tapsrc, int32_t ntap, int32_t CoP, int32_t CiP,
                    int32_t Cout, int32_t Cin, int64_t sCo, int64_t sCi, float* dw,
                    const double* dbias_sums /* or NULL */, float* dbias_grad /* or NULL */, int32_t nbias,
                    int32_t dbias_stride /* as above */, sp_stream_t stream);

/* ------------------------------------------------------------------ first layer, read from the NCDHW fp32 input
 * (Unet3D.py:18-20 of block1: BatchNorm3d(2) -> Conv3d(2,16,3) -> LeakyReLU; train_unet_segmentation.py:22).  With two
 * input channels the im2col K dimension is packed (9 (dz,dy) groups x (4 dx x 2 channels)) instead of padding the
 * channels to a 16-wide plane; bf16 storage only.  sp_first_supported tells whether a layer shape has this path. */
int sp_first_supported(int32_t Cin, int32_t Cout, int32_t k);
/* sums[rep][CP][2] += (sum x, sum x^2) per input channel of x [B][C][DHW] (values rounded to bf16 first) */
int sp_bn_stats_ncdhw(const float* x, int32_t B, int32_t C, int64_t DHW, int32_t CP, double* sums, int32_t nrep,
                      sp_stream_t stream);
/* w (16,2,3,3,3), b (16), BatchNorm scale/shift (2) or NULL -> wfrag (3*64*8 bf16 MFMA fragments), bias_f (16) */
int sp_first_prep(const float* w, const float* b, const float* scale, const float* shift, void* wfrag, float* bias_f,
                  sp_stream_t stream);
/* y [B][D-2][H-2][W-2][16] bf16 = act(conv + bias); stats[rep][16][2] += (sum y, sum y^2) when stats != NULL */
int sp_first_conv_fwd(const float* x, int32_t B, int32_t D, int32_t H, int32_t W, const void* wfrag, const float* bias_f,
                      int32_t act, float act_param, void* y, double* stats, int32_t nrep, sp_stream_t stream);
/* partials [nblocks][27][16][2] fp32 (written): raw-input weight gradient blocks for
 * sp_wgrad_finish_folded(partials, nblocks, tapsrc, 27, 16, 2, 16, 2, ...) */
int sp_first_wgrad(const float* x, const void* dz, int32_t B, int32_t D, int32_t H, int32_t W, float* partials,
                   int32_t nblocks, sp_stream_t stream);
/* the same with dz formed on the fly: dz = (coef[0][c]*g + coef[1][c]*y + coef[2][c]) * act'(y) (coef: [3][16], the
 * BatchNorm-backward triple of the NEXT layer; g its data gradient, y this layer's output; both [..][16] bf16), and
 * dbias_sums[c] (fp64, zeroed by the caller) += sum dz -- replaces sp_bn_act_bwd + sp_first_wgrad for this layer */
int sp_first_wgrad_fused(const float* x, const void* g, const void* y, const float* coef, int32_t act, float act_param,
                         int32_t B, int32_t D, int32_t H, int32_t W, float* partials, int32_t nblocks, double* dbias_sums,
                         sp_stream_t stream);
/* The same four for Cout = 16 or 32 output channels (Conv3d(2, 32, 3): the first layer of the 4-scale network, BASELINE.json
 * configs[4]): every "16" above reads Cout -- wfrag (Cout/16)*3*64*8 bf16, bias_f / y rows / coef rows / dbias rows of Cout,
 * partials [nblocks][27][Cout][2].  sp_first_conv_fwd_n can also write y8, the e4m3 plane-major copy of y
 * ([Cout/16][B][D-2][H-2][W-2][16 bytes], y8_plane bytes per plane; NULL: none) for an fp8 second layer. */
int sp_first_prep_n(const float* w, const float* b, const float* scale, const float* shift, void* wfrag, float* bias_f,
                    int32_t Cout, sp_stream_t stream);
int sp_first_conv_fwd_n(const float* x, int32_t B, int32_t D, int32_t H, int32_t W, const void* wfrag, const float* bias_f,
                        int32_t act, float act_param, void* y, double* stats, int32_t nrep, int32_t Cout, void* y8,
                        int64_t y8_plane, sp_stream_t stream);
int sp_first_wgrad_n(const float* x, const void* dz, int32_t B, int32_t D, int32_t H, int32_t W, float* partials,
                     int32_t nblocks, int32_t Cout, sp_stream_t stream);
int sp_first_wgrad_fused_n(const float* x, const void* g, const void* y, const float* coef, int32_t act, float act_param,
                           int32_t B, int32_t D, int32_t H, int32_t W, float* partials, int32_t nblocks, double* dbias_sums,
                           int32_t Cout, sp_stream_t stream);
/* the same with y as its e4m3 plane-major copy (what sp_first_conv_fwd_n wrote to y8 / y8_plane; the "fp8" precision mode stores no
 * 16-bit y of the first layer: y = NULL there) -- runtime/layers.py:FirstConvLayer.backward */
int sp_first_wgrad_fused_y8(const float* x, const void* g, const void* y8, int64_t y8_plane, const float* coef, int32_t act,
                            float act_param, int32_t B, int32_t D, int32_t H, int32_t W, float* partials, int32_t nblocks,
                            double* dbias_sums, int32_t Cout, sp_stream_t stream);

/* ------------------------------------------------------------------ layout
 * NCDHW fp32 (reference layout, README.md:13 / data.py:305) <-> channels-last-3d */
int sp_ncdhw_to_cl(const float* src, void* dst, int32_t dtype, int32_t B, int32_t C, int64_t DHW, int32_t CP,
                   sp_stream_t stream);
int sp_cl_to_ncdhw(const void* src, float* dst, int32_t dtype, int32_t B, int32_t C, int64_t DHW, int32_t CP,
                   sp_stream_t stream);

/* ------------------------------------------------------------------ BatchNorm3d pieces (Unet3D.py:18,21; Cae3D.py:40..217)
 * sums[c] = (sum x, sum x^2) over all nvox voxels of a channels-last tensor */
int sp_bn_stats(const void* x, int32_t dtype, int64_t nvox, int32_t CP, double* sums /* [CP][2] */,
                sp_stream_t stream);
/* sums -> scale/shift for BN-on-load; updates running stats (momentum, unbiased var) like
 * nn.BatchNorm3d; writes mean/invstd for the backward.  training=0: use running stats. */
int sp_bn_finalize(const double* sums /* [nrep][CP][2], replicas are added */, int32_t nrep, double count,
                   const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, int32_t training, int32_t C, int32_t CP,
                   float* scale, float* shift, float* mean, float* invstd, sp_stream_t stream);
/* backward reductions: sums[c] = (sum g, sum g*x) over the tensor */
int sp_bn_bwd_reduce(const void* g, const void* x, int32_t dtype, int64_t nvox, int32_t CP, double* sums,
                     sp_stream_t stream);
/* from (sum g, sum g*x): dgamma, dbeta (ACCUMULATED into the gradient buffers, may be NULL) and
 * coef[3][CP] with dx = coef0*g + coef1*x + coef2 */
int sp_bn_bwd_finalize(const double* sums /* [nrep][CP][2] */, int32_t nrep, double count, const float* gamma, const float* mean,
                       const float* invstd, int32_t C, int32_t CP, float* dgamma, float* dbeta, float* coef,
                       float param_grad_scale /* dgamma/dbeta += scale * value; 1/world when the sums are global */,
                       sp_stream_t stream);
/* dz = (coef0*g + coef1*y + coef2) * act'(y)  (coef NULL: dz = g*act'(y));
 * dbias_sums[c] += sum_voxels dz (fp64, may be NULL) */
int sp_bn_act_bwd(const void* g, const void* y, const float* coef, int32_t dtype, int64_t nvox, int32_t CP,
                  int32_t act, float act_param, void* dz, double* dbias_sums, sp_stream_t stream);

/* The same three for G BatchNorm GROUPS in one launch -- the batched passes of the CAE (Cae3D.py:105-107 three encoder calls,
 * :230-233 four decoder calls): the passes are stacked along the batch axis (samples [g*Bg, (g+1)*Bg) = pass g), every pass
 * keeps its OWN batch statistics, and the running statistics take the G momentum updates in pass order.  sums
 * [G][nrep][CP][2]; scale / shift of group g at scale + g*coef_stride (rows 0 / 2 of a [G][3][CP] table: coef_stride = 3*CP);
 * mean / invstd [G][CP]; coef [G][3][CP].  sp_bn_act_bwd_groups: voxels [g*group_vox, (g+1)*group_vox) use coef + g*3*CP. */
int sp_bn_finalize_groups(const double* sums, int32_t nrep, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, int32_t training, int32_t C,
                          int32_t CP, int32_t G, int32_t coef_stride, float* scale, float* shift, float* mean, float* invstd,
                          sp_stream_t stream);
int sp_bn_bwd_finalize_groups(const double* sums, int32_t nrep, double count, const float* gamma, const float* mean,
                              const float* invstd, int32_t C, int32_t CP, int32_t G, float* dgamma, float* dbeta, float* coef,
                              float param_grad_scale, sp_stream_t stream);
int sp_bn_act_bwd_groups(const void* g, const void* y, const float* coef, int32_t dtype, int64_t nvox, int32_t CP, int32_t act,
                         float act_param, void* dz, double* dbias_sums, int64_t group_vox, sp_stream_t stream);
/* sp_bn_act_bwd_groups for a layer whose weight gradient reads the RAW layer input although its BatchNorm differs per group and
 * its convolution pads (the CAE's batched passes with the BatchNorm folded per group, sp_conv_args.bias_tab): next to dz
 * [B][D][H][W][CP] it leaves the sums of the stored dz over the border classes of that output grid -- (2 pad + 1) classes per
 * axis, class = (cz * ny + cy) * nx + cx -- per group in cls_sums[G][ncls][CP] (fp64, zeroed by the caller; <= 75 classes). */
int sp_bn_act_bwd_groups_cls(const void* g, const void* y, const float* coef, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W,
                             int32_t CP, int32_t act, float act_param, void* dz, double* dbias_sums, int32_t group_batch,
                             int32_t padD, int32_t padH, int32_t padW, double* cls_sums, sp_stream_t stream);
/* ... and the finish of that layer's weight gradient (sp_conv3d_wgrad on the raw input, row-sliding kernel: partial block i
 * belongs to group i * G / nparts): dw[co][ci][tap] += sum_g s_g[ci] A_g + t_g[ci] Sv_g[tap][co] with Sv_g the class sums whose tap
 * lies inside the input, dbias_grad[co] += sum dz, and the BatchNorm-backward pair (sum g, sum g x) of the layer's input per group
 * into bn_sums[G][bn_nrep][bn_cp][2] (NULL: skip) -- the data gradient then needs no statistics epilogue.  coef: rows
 * (scale, -, shift) of pitch coef_pitch per group. */
int sp_wgrad_finish_folded_groups(const float* dw_acc, int32_t nparts, int32_t ntap /* 27, or 1 (pointwise: padding 0) */, int32_t G, int32_t CoP, int32_t CiP, int32_t Cout, int32_t Cin,
                                  int64_t sCo, int64_t sCi, const float* coef, int32_t coef_gstride, int32_t coef_pitch,
                                  const double* cls_sums, int32_t padD, int32_t padH, int32_t padW, const float* w, float* dw,
                                  float* dbias_grad, double* bn_sums, int32_t bn_nrep, int32_t bn_cp, sp_stream_t stream);

/* The CAE decoder's output layer BatchNorm3d(n <= 16) -> Conv3d(n, 1, 1) -> Sigmoid (Cae3D.py:214-218) as streaming kernels
 * (csrc/sp_pwout.hip; bf16 channels-last input of pitch 16, NCDHW fp32 output of one channel):
 * forward straight from the RAW input -- the BatchNorm's rows (scale, -, shift) of pitch 16 per group (coef, coef_gstride floats
 * apart; group of sample b = b / group_batch, 0: one group; NULL: no BatchNorm) folded into 16 coefficients;
 * backward: g[b, v, c] = w_c dz with dz = dout out (1 - out) -- the gradient at the BatchNorm's output -- and, per group, the 17
 * sums (sum dz, sum dz x_c) in nrep replica rows of 32 doubles (zeroed by the caller);
 * finish: the BatchNorm-backward pair (sum g_c, sum g_c x_c) into replica row 0 of bn_sums ([G][bn_nrep][16][2], zeroed by the
 * caller; NULL: skip), dw[c] += s_c sum dz x_c + t_c sum dz and dbias += sum dz (NULL: a frozen layer). */
int sp_pwout_fwd(const void* x, int32_t B, int64_t V, int32_t Cin, int32_t CP, const float* coef, int32_t coef_gstride,
                 int32_t group_batch, const float* w, const float* bias, float* out, sp_stream_t stream);
int sp_pwout_bwd(const float* dout, const float* out, const void* x, int32_t B, int64_t V, int32_t Cin, int32_t CP, const float* w,
                 int32_t group_batch, int32_t nrep, void* g, double* sums, sp_stream_t stream);
int sp_pwout_finish(const double* sums, int32_t nrep, int32_t G, int32_t Cin, const float* w, const float* coef, int32_t coef_gstride,
                    double* bn_sums, int32_t bn_nrep, float* dw, float* dbias, sp_stream_t stream);

/* ------------------------------------------------------------------ pooling / upsampling / skip (Unet3D.py:59-72)
 * MaxPool3d(2,2) floor mode; optional output statistics [CP][2] */
int sp_maxpool2_fwd(const void* x, void* y, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP,
                    double* stats, sp_stream_t stream);
/* trilinear x2 (align_corners=0) into channels [0,CP) of a wider tensor with pitch CPd */
int sp_upsample2_fwd(const void* x, void* y, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP,
                     int32_t CPd, double* stats, sp_stream_t stream);
/* centre-crop copy of src (pitch CPs) into channels [c0, c0+CPs) of dst (pitch CPd); stats [CPs][2] */
int sp_crop_copy(const void* src, void* dst, int32_t dtype, int32_t B, int32_t Ds, int32_t Hs, int32_t Ws,
                 int32_t CPs, int32_t Dd, int32_t Hd, int32_t Wd, int32_t CPd, int32_t c0, double* stats,
                 sp_stream_t stream);
/* both of the above in one pass over the concat buffer (full-line writes): cat[..., 0:CPu) = upsample2(low),
 * cat[..., CPu:CPu+CPs) = centre crop of skip; stats[c][2] over all CPd = CPu + CPs channels (Unet3D.py:67-72) */
int sp_upsample2_crop_cat_fwd(const void* low, int32_t CPu, const void* skip, int32_t CPs, void* cat, int32_t CPd,
                              int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds, int32_t Hs, int32_t Ws,
                              int64_t cat_plane /* != 0: plane-major output, elements per 16-channel plane */, double* stats,
                              sp_stream_t stream);
/* gradient of a block output y that feeds (a) MaxPool3d(2,2) -> BN -> ... and (b) the cropped skip:
 * dz = [ poolbwd(coefp0*gp + coefp1*pool(y) + coefp2) + crop-region(coefs0*gs + coefs1*cat + coefs2) ] * act'(y)
 * either source may be NULL */
/* (gs, cs0, CPcat): the skip gradient is channels [cs0, cs0+CP) of a tensor with channel pitch CPcat -- the concat
 * gradient itself, or a dense tensor holding only the skip part (cs0 = 0, CPcat = CP).  (coefs, coef_c0, coef_stride):
 * its BatchNorm-backward triple is coefs[k*coef_stride + coef_c0 + c]; coef_stride <= 0: (cs0, CPcat) apply.  cat is
 * unused (the skip half of the concat buffer is a copy of y) and may be NULL. */
int sp_pool_skip_act_bwd(const void* y, const void* gp, const float* coefp, const void* cat, const void* gs,
                         const float* coefs, int32_t cs0, int32_t CPcat, int32_t coef_c0, int32_t coef_stride,
                         int32_t dtype, int32_t B, int32_t D,
                         int32_t H, int32_t W, int32_t CP, int32_t Dc, int32_t Hc, int32_t Wc, int32_t act,
                         float act_param, void* dz, double* dbias_sums, sp_stream_t stream);
/* gradient through trilinear x2: dz[lowres] = upsample^T(coef0*g + coef1*cat + coef2)[channels 0..CP) * act'(y) */
/* g: channel pitch CPcat, the upsampled part in channels [0, CP); coef[k*coef_stride + c] (coef_stride <= 0: CPcat).
 * cat (the concat buffer, same pitch as g) is only read by the gather fallback (channel counts whose octets do not
 * divide 256); NULL otherwise. */
int sp_upsample2_act_bwd(const void* y, const void* cat, const void* g, const float* coef, int32_t CPcat,
                         int32_t coef_stride, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, int32_t act,
                         float act_param, void* dz, double* dbias_sums, sp_stream_t stream);
/* the same with the fp8 plane-major copy of dz (as sp_bn_act_bwd_q8; dz == NULL: only the copy) */
int sp_upsample2_act_bwd_q8(const void* y, const void* cat, const void* g, const float* coef, int32_t CPcat, int32_t coef_stride,
                            int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, int32_t act, float act_param,
                            void* dz, double* dbias_sums, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale,
                            sp_stream_t stream);

/* ------------------------------------------------------------------ the same kernels with an fp8 shadow output
 * ("fp8" precision mode): besides the bf16 tensor the kernel writes q8 = fp8(q8_scale * stored value) PLANE-MAJOR,
 * [CP/16][voxels][16 bytes] with q8_plane bytes per plane -- the operand of the next sp_conv3d_zm8 (q8_fmt 0 = e4m3: the
 * input of a forward convolution; 1 = e5m2: the output gradient read by a data gradient) -- instead of a separate
 * sp_quantize_f8 pass over HBM.  bf16 tensors with CP % 16 == 0 only; every other argument as in the plain entry point.
 * sp_upsample2_crop_cat_fwd_q8 needs the plane-major concat (cat_plane != 0, CPu and CPs multiples of 16). */
int sp_bn_act_bwd_q8(const void* g, const void* y, const float* coef, int32_t dtype, int64_t nvox, int32_t CP, int32_t act,
                     float act_param, void* dz, double* dbias_sums, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale,
                     sp_stream_t stream);
/* ... with y given as its e4m3 plane-major copy ([CP/16][nvox][16 bytes], y8_plane bytes per plane): the "fp8" precision mode
 * stores no 16-bit output for a layer all of whose readers take the copy (sp_conv3d_zm8 with y = NULL); q8 may be NULL */
int sp_bn_act_bwd_y8(const void* g, const void* y8, int64_t y8_plane, const float* coef, int64_t nvox, int32_t CP, int32_t act,
                     float act_param, void* dz, double* dbias_sums, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale,
                     sp_stream_t stream);
int sp_maxpool2_fwd_q8(const void* x, void* y, int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP,
                       double* stats, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream);
/* ... with the input as its e4m3 plane-major copy (the "fp8" mode stores no 16-bit output for a convolution all of whose readers take
 * the copy); y may be NULL (only the e4m3 copy of the pooled tensor is wanted) -- runtime/unet_engine.py */
int sp_maxpool2_fwd_x8(const void* x8, int64_t x8_plane, void* y, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, double* stats,
                       void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream);
int sp_upsample2_crop_cat_fwd_q8(const void* low, int32_t CPu, const void* skip, int32_t CPs, void* cat, int32_t CPd,
                                 int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds, int32_t Hs, int32_t Ws,
                                 int64_t cat_plane, double* stats, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale,
                                 sp_stream_t stream);
/* ... with the skip tensor as its e4m3 plane-major copy ([CPs/16][B][Ds][Hs][Ws][16 bytes]); plane-major output only */
int sp_upsample2_crop_cat_fwd_q8s8(const void* low, int32_t CPu, const void* skip8, int64_t skip8_plane, int32_t CPs, void* cat, int32_t CPd,
                                   int32_t B, int32_t D, int32_t H, int32_t W, int32_t Ds, int32_t Hs, int32_t Ws, int64_t cat_plane,
                                   double* stats, void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream);
int sp_pool_skip_act_bwd_q8(const void* y, const void* gp, const float* coefp, const void* cat, const void* gs,
                            const float* coefs, int32_t cs0, int32_t CPcat, int32_t coef_c0, int32_t coef_stride,
                            int32_t dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t CP, int32_t Dc, int32_t Hc,
                            int32_t Wc, int32_t act, float act_param, void* dz, double* dbias_sums, void* q8, int64_t q8_plane,
                            int32_t q8_fmt, float q8_scale, sp_stream_t stream);
/* ... with y as its e4m3 plane-major copy; q8 may be NULL */
int sp_pool_skip_act_bwd_y8(const void* y8, int64_t y8_plane, const void* gp, const float* coefp, const void* gs, const float* coefs,
                            int32_t cs0, int32_t CPcat, int32_t coef_c0, int32_t coef_stride, int32_t B, int32_t D, int32_t H, int32_t W,
                            int32_t CP, int32_t Dc, int32_t Hc, int32_t Wc, int32_t act, float act_param, void* dz, double* dbias_sums,
                            void* q8, int64_t q8_plane, int32_t q8_fmt, float q8_scale, sp_stream_t stream);

/* ------------------------------------------------------------------ network output side + Dice (Unet3D.py:53,75-77;
 * metrics.py:16-28).  dz[b,v,c] = dout[b,c,v]*act'(out[b,c,v]) : NCDHW fp32 -> channels-last */
int sp_out_grad_to_cl(const float* dout, const float* out, int32_t B, int32_t C, int64_t DHW, int32_t CP,
                      int32_t dtype, int32_t act, float act_param, void* dz, double* dbias_sums,
                      sp_stream_t stream);
/* ------------------------------------------------------------------ BatchDiceLoss (metrics.py:16-28)
 * o, t: (B, C, DHW) fp32, contiguous per sample, arbitrary batch stride (elements) so that channel-slice views
 * (dto.outputs.core / .penu, UnetDto) are read in place.  sums[row][3*c + k] (fp64, zeroed by the caller;
 * SP_REDUCE_ROWS replica rows of SP_DICE_PITCH(C) doubles, added up by sp_dice_finalize) +=
 * (sum o*t, sum o*o, sum t*t) over batch and volume. */
int sp_dice_sums(const float* o, int64_t o_bstride, const float* t, int64_t t_bstride, int32_t B, int32_t C, int64_t DHW,
                 double* sums, sp_stream_t stream);
/* loss = 1 - sum_c w_c (2 I_c + eps)/(O_c + T_c + eps);  coef[c] = (ca, cb): d loss/d o = ca*t + cb*o */
int sp_dice_finalize(const double* sums, const float* weights, double eps, int32_t C, float* loss, float* coef,
                     sp_stream_t stream);
/* the same, and the replica rows are zeroed again after they are read: the caller keeps one accumulator across steps and needs no fill
 * launch in front of the next sp_dice_sums (BatchDiceLoss of common/metrics.py in a captured training step) */
int sp_dice_finalize_clear(double* sums, const float* weights, double eps, int32_t C, float* loss, float* coef,
                           sp_stream_t stream);
/* The CAE reconstruction loss (CaeReconstructionLearner.py:52-70) in three launches:
 *   [ mean(|p-i| - (p-i)) + mean(|p-c| - (p-c)) + Dice(c, tc) + Dice(p, tp) + Dice(l, tl) + factor * mean|zi - zl| ] / (5 + factor)
 * c, p, l, i: the reconstructions (B, 1, DHW) fp32 with batch strides *bs (elements; slices of a stacked tensor are read in place),
 * t*: the ground truths, zi / zl: the latents (nlat elements each, dense).  Dice = 1 - w (2 I + eps) / (O + T + eps) over batch and
 * volume (metrics.py:16-28, one class).  sums: SP_REDUCE_ROWS x 16 doubles, zeroed by the caller; coef: 8 floats for the backward.
 * sp_cae_loss_bwd: dense (B, DHW) gradients of c, p, l, i and (nlat) of zi, zl, times the upstream scalar *up (device memory). */
int sp_cae_loss_fwd(const float* c, int64_t cbs, const float* p, int64_t pbs, const float* l, int64_t lbs, const float* i, int64_t ibs,
                    const float* tc, int64_t tcbs, const float* tp, int64_t tpbs, const float* tl, int64_t tlbs, int32_t B, int64_t DHW,
                    const float* zi, const float* zl, int64_t nlat, float dice_weight, double eps, float factor, double* sums,
                    float* loss, float* coef, sp_stream_t stream);
int sp_cae_loss_bwd(const float* c, int64_t cbs, const float* p, int64_t pbs, const float* l, int64_t lbs, const float* i, int64_t ibs,
                    const float* tc, int64_t tcbs, const float* tp, int64_t tpbs, const float* tl, int64_t tlbs, int32_t B, int64_t DHW,
                    const float* coef, const float* up, float* dc, float* dp, float* dl, float* di, const float* zi, const float* zl,
                    int64_t nlat, float* dzi, float* dzl, sp_stream_t stream);
/* dout (contiguous) = *upstream * (ca[c]*t + cb[c]*o); upstream: device pointer to the scalar gradient (NULL = 1) */
int sp_dice_bwd(const float* o, int64_t o_bstride, const float* t, int64_t t_bstride, const float* coef,
                const float* upstream, int32_t B, int32_t C, int64_t DHW, float* dout, sp_stream_t stream);

/* ------------------------------------------------------------------ fused classify head (Unet3D.py:49-54,75-77)
 * seg = sigmoid(W2 * lrelu(W1*x + b1) + b2): x channels-last [B*nvox][CP], seg NCDHW fp32 [B][NC][nvox].
 * w1 is (CH, C), w2 is (NC, CH) row-major (the Conv3d 1x1x1 weights).  sp_head_supported lists the shapes with a
 * fused kernel; other shapes run as two generic sp_conv3d_igemm layers. */
int sp_head_supported(int32_t C, int32_t CH, int32_t NC);
/* the same with the storage type: (C, CH, NC) = (32, 32, 2), the head of the 4-scale network, exists for SP_BF16 only */
int sp_head_supported_dtype(int32_t C, int32_t CH, int32_t NC, int32_t dtype);
int sp_head_fwd(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                const float* b1, int32_t CH, const float* w2, const float* b2, int32_t NC, float slope, float* seg,
                sp_stream_t stream);
/* backward: dz = dL/dx * act_x'(x) (x is the producing conv's post-activation output).  Each workgroup WRITES one
 * row of partial sums [dW1 (CH*C) | db1 (CH) | dW2 (NC*CH) | db2 (NC) | sum dz (C)] to partials
 * (sp_head_bwd_rows(B*nvox_per_b) rows of sp_head_row_floats(C,CH,NC) floats; no atomics, run-to-run
 * deterministic); sp_head_grad_finish adds the rows into the four parameter gradients (+=) and into the producing
 * conv's bias-gradient sums. */
int64_t sp_head_bwd_rows(int64_t total_voxels);
int32_t sp_head_row_floats(int32_t C, int32_t CH, int32_t NC);
int sp_head_bwd(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                const float* b1, int32_t CH, const float* w2, int32_t NC, float slope, const float* seg,
                const float* dseg, int32_t act_x, float act_x_param, void* dz, float* partials, sp_stream_t stream);
/* the same with the fp8 plane-major copy of dz ([C/16][B*nvox][16 bytes], q8_fmt 0 = e4m3 / 1 = e5m2 of q8_scale * dz, rounded
 * from the stored 16-bit value) for an fp8 data / weight gradient of the producing layer; dz == NULL: only the copy */
int sp_head_bwd_q8(const void* x, int32_t dtype, int64_t nvox_per_b, int32_t B, int32_t CP, int32_t C, const float* w1,
                   const float* b1, int32_t CH, const float* w2, int32_t NC, float slope, const float* seg, const float* dseg,
                   int32_t act_x, float act_x_param, void* dz, float* partials, void* q8, int64_t q8_plane, int32_t q8_fmt,
                   float q8_scale, sp_stream_t stream);
int sp_head_grad_finish(const float* partials, int64_t rows, int32_t C, int32_t CH, int32_t NC, float* gW1, float* gb1,
                        float* gW2, float* gb2, double* dbias_sums, sp_stream_t stream);

/* ------------------------------------------------------------------ batch metrics (metrics.py:31-62)
 * counts[4] (zeroed by the caller) += tp, fp, fn, tn of (result > threshold) vs (target > threshold), n fp32 elements */
int sp_confusion_counts(const float* result, const float* target, float threshold, int64_t n,
                        unsigned long long* counts, sp_stream_t stream);

/* ------------------------------------------------------------------ small utilities */
int sp_add_f64_to_f32(const double* src, float* dst, int64_t n, float scale, sp_stream_t stream); /* dst += scale*src */
int sp_axpby(const void* x, const void* y, void* out, int32_t dtype, int64_t n, float a, float b, sp_stream_t stream);
/* latent lerp Cae3D.py:78-89: out[b,...] = c + step[b]*(p - c) */
int sp_lerp_batch(const void* c, const void* p, const float* step, void* out, int32_t dtype, int32_t B,
                  int64_t per_b, sp_stream_t stream);

/* ------------------------------------------------------------------ optimiser (train_unet_segmentation.py:32,
 * train_shape_reconstruction.py:40): torch.optim.Adam semantics (L2-coupled weight decay, bias
 * correction, no amsgrad) on flat fp32 buffers; g is multiplied by grad_scale first */
int sp_adam_step_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int32_t step, float grad_scale, sp_stream_t stream);

/* same, with the 1-based step count read from device memory (hipGraph-capturable training step) */
int sp_adam_step_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, const int32_t* step_dev, float grad_scale,
                          sp_stream_t stream);

/* same, with {lr, beta1, beta2, eps, weight_decay} read from device memory too: a captured step keeps following
 * Learner.adapt_lr (learner/Learner.py:156-158, MultiStepLR) and adapt_betas (CaeReconstructionLearner.py:28-40) */
int sp_adam_step_flat_hyp(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                          const int32_t* step_dev, float grad_scale, sp_stream_t stream);

/* ------------------------------------------------------------------ data-parallel gradient exchange (RCCL over xGMI)
 * The reference has no distributed code; the path shards by batch and needs ONE collective per step, the sum of the flat
 * fp32 gradient buffer over the replicas (learner/Learner.py:120-122 per replica).  comm is an ncclComm_t of
 * the process's librccl (resolved at run time: these entry points fail with SP_EHIP where RCCL is absent, the rest of the
 * library does not depend on it).  Host side: rank 0 calls sp_comm_unique_id, ships the 128 bytes to the other ranks (any
 * side channel: torch.distributed's store, MPI, a file), every rank calls sp_comm_init_rank with its device current.
 * Every collective is enqueued on the caller's stream and returns. */
int sp_comm_available(void);                                   /* 1 when librccl and its symbols were found */
int sp_comm_unique_id(void* id128);                            /* ncclGetUniqueId -> 128 bytes */
int sp_comm_init_rank(void** comm, int32_t nranks, const void* id128, int32_t rank);
int sp_comm_destroy(void* comm);
int sp_allreduce_flat(void* comm, float* buf, int64_t n, sp_stream_t stream);        /* buf = sum over ranks, in place */
/* the same for the fp64 accumulators the exact data-parallel mode exchanges (BatchNorm sum / sum^2 and sum g / sum g*x rows, Dice
 * sums: SURVEY 8e items 2-3); a few KB per call, latency-bound */
int sp_allreduce_flat_f64(void* comm, double* buf, int64_t n, sp_stream_t stream);
/* two-shot form for the xGMI full mesh: buf holds nranks chunks of `chunk` floats; after the reduce-scatter chunk `rank` of
 * this rank's buf is the sum; the all-gather then fills the other chunks */
int sp_reduce_scatter_flat(void* comm, float* buf, int64_t chunk, int32_t rank, sp_stream_t stream);
int sp_allgather_flat(void* comm, float* buf, int64_t chunk, int32_t rank, sp_stream_t stream);

/* ------------------------------------------------------------------ input pipeline (SURVEY.md 8 "next" row N4)
 * ElasticDeform.elastic_transform (common/data.py:326-339) on the device.  Volumes are C-ordered (n0, n1, n2) fp32
 * arrays, the reference's (x, y, z) numpy layout.
 * sp_gaussian_filter3d = scipy.ndimage.gaussian_filter(src, sigma, mode="constant", cval=0, truncate) (data.py:332-334):
 * radius int(truncate*sigma + 0.5) (<= 64), normalised weights, axis 0 then 1 then 2; tmp = scratch of the same size;
 * src, dst, tmp: three different buffers. */
int sp_gaussian_filter3d(const float* src, float* dst, float* tmp, int32_t n0, int32_t n1, int32_t n2, float sigma,
                         float truncate, sp_stream_t stream);
/* sp_map_coordinates_linear = scipy.ndimage.map_coordinates(image, (i + s0*d0, j + s1*d1, k + s2*d2), order=1,
 * mode="constant", cval) (data.py:336-339): cval wherever a coordinate leaves [0, n-1], else trilinear interpolation */
int sp_map_coordinates_linear(const float* image, const float* d0, const float* d1, const float* d2, float s0, float s1, float s2,
                              float cval, float* out, int32_t n0, int32_t n1, int32_t n2, sp_stream_t stream);

/* ------------------------------------------------------------------ surface distances of the batch metrics
 * metrics.py:42-44 -> medpy 0.3.0 metric.binary.hd / assd (__surface_distances): border = mask XOR binary_erosion(mask)
 * with the cross structure of the array's rank (out-of-bounds = background), exact Euclidean distance transform of the
 * complement of the other mask's border (unit spacing), sampled at this mask's border.  result / reference: fp32 arrays
 * of rank ndim <= 5 (the reference passes the whole (B,1,D,H,W) tensors: the transform then runs across the batch axis
 * too, and the extent-1 channel axis makes every voxel a border voxel -- reproduced), mask = value > threshold.
 * ws: 4*prod(dims) floats; out[6] (fp64, zeroed by the caller) = {max SQUARED distance (an exact integer), sum of the
 * distances, count} result-border -> reference-border, then the same for reference -> result:
 * hd = sqrt(max(out[0], out[3])),  assd = (out[1]/out[2] + out[4]/out[5]) / 2. */
int sp_surface_distances(const float* result, const float* reference, float threshold, int32_t ndim, const int32_t* dims,
                         float* ws, double* out, sp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STROKE_AMD_H */
