// Output-stationary z-marching convolution for the stride-1 3x3x3 layers of the U-Net (bf16, LDS-DMA staging).
//
// Replaces nn.Conv3d(3, padding 0) forward and its data gradient (Unet3D.py:19,22) on the layers whose weight
// fragments fit the register file: Cin = 16 P, Cout = 16 NT with 3 * ceil(18 P / 4) * NT <= ~56 fragments.
//
// What the tiled kernel (sp_conv_dma.hip, conv_igemm_dma_kernel) and the ring kernel (conv_igemm_zs_kernel) leave on the
// table, by their own phase stamps: stage -> K loop -> store run one after the other inside a workgroup, every input
// plane is staged 2-3 times (z halo), and with Cout = 16 every activation fragment read from LDS feeds ONE MFMA.
// Here a workgroup owns a COLUMN of th x tw output voxels (<= 16 NW MT of them, flattened row-major onto the MFMA column
// groups: ConvZmDev.tw / th) and marches through the INPUT planes of that column:
//
//   * an input plane zi contributes to the three output planes zi, zi-1, zi-2 (taps dz = 0, 1, 2).  The wave keeps the
//     accumulators of all three in registers, so a plane is staged ONCE, lives in LDS for exactly one step, and every
//     activation fragment read from LDS feeds 3 x NT MFMAs (in-plane taps (dy, dx) x channels are the K loop: 18 P
//     octets = ceil(18 P / 4) steps of 32);
//   * the LDS ring therefore holds only planes in flight (2 or 3 slots): the DMA of plane zi+1 (zi+2) is issued at the
//     top of step zi and has a whole step of MFMAs to land; one barrier per step;
//   * all weight fragments (3 dz x KS x NT) are resident in registers -- one workgroup per CU, 512 registers per lane;
//   * the epilogue of a finished plane (bias, activation, bf16 pack, BatchNorm statistics, stores) is carried into the
//     NEXT step's instruction stream, where its VALU work issues between that step's MFMAs; stores are issued a whole
//     step before the wait that has to cover them (vmcnt counts stores too on gfx9);
//   * padding (data gradients: "full" correlation) costs nothing extra: out-of-volume 16-byte chunks are DMA'd from a
//     zero page, so every wave issues the same number of DMA instructions per plane and the waits can be counted.
//
// K table (host, runtime/plan.py): ktab[s * 4 + g] = byte offset inside a slot of the octet lane group g reads in step s
// (plane p * ITH * 18 + dy * (tw + 2) + dx) * 32 + octet * 16 (a plane keeps its compile-time pitch of ITH x 18 voxels); weight fragments: [(dz * KS + s) * NT + n] * 64 lanes, packed by
// sp_conv_prep_weights / sp_conv_prep_folded from the matching kmap (the BatchNorm of the un-padded forward convolution is
// folded into weights and bias there, as for every DMA kernel).
#include "sp_common.h"
#include <string.h>

struct ConvZmDev {
  sp_conv_args a;
  const void* zeros;      // >= 16 readable zero bytes
  int32_t nty, ntx;
  uint32_t ncols;
  FastDiv d_tx, d_ty;
  // the workgroup's output tile: th rows of tw voxels, tw * th <= 16 NW MT.  The 16 voxels of an MFMA column group are 16
  // CONSECUTIVE voxels of the flattened (row, column) index, so a tile need not be 16 wide: rows of 50 run as 25 x 10 tiles
  // (250 of 256 slots used) instead of four 16-wide tiles per row (a.TH / a.ITW; 0: the classic NW MT rows x 16)
  int32_t tw, th, itw;
  FastDiv d_tw, d_itw;
};

// ---- diagnostic build (-DSP_ZM_STAMPS, tools/stamp_zm.py): cycles per step segment, summed per wave -----------------
#ifdef SP_ZM_STAMPS
__device__ unsigned long long sp_zm_stamp_buf[1024][8][8];
#define ZM_T(var)                                                                       \
  unsigned long long var;                                                               \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#define ZM_ACC(k, t1, t0) zm_sum[k] += (t1) - (t0)
extern "C" int sp_debug_zm_stamps(void* out, int nbytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sp_zm_stamp_buf), nbytes) == hipSuccess ? 0 : -2;
}
#else
#define ZM_T(var)
#define ZM_ACC(k, t1, t0)
#endif

typedef unsigned int zm_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int zm_u32x4 __attribute__((ext_vector_type(4)));

// four consecutive output channels of one voxel through a buffer descriptor: an out-of-range byte offset DROPS the store in
// hardware, so rows / columns / planes outside the output need no branch (branches would cut the epilogue out of the
// basic block whose MFMAs it is meant to issue between)
template <typename T> struct ZmStore;
typedef sp_h16 zm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float zm_f32x2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32; element-wise casts packed by hand
// cost a conversion, a shift and an or per element
__device__ __forceinline__ uint32_t zm_pack2(float lo, float hi) {
  const zm_f32x2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, zm_bf16x2));
}
// four floats -> four e4m3 bytes (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint32_t zm_pack4_e4m3(const float* v, float s) {
  const float a = __builtin_amdgcn_fmed3f(v[0] * s, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(v[1] * s, -448.f, 448.f);
  const float c = __builtin_amdgcn_fmed3f(v[2] * s, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(v[3] * s, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
template <> struct ZmStore<bf16_t> {
  static __device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, uint32_t off, const float* v) {
    zm_u32x2 d = {zm_pack2(v[0], v[1]), zm_pack2(v[2], v[3])};
    __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, 0);
  }
};
template <> struct ZmStore<float> {
  static __device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, uint32_t off, const float* v) {
    zm_u32x4 d = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, 0);
  }
};

// wait for the wave's own DMA (all but the N youngest vector-memory operations), then the workgroup barrier: a raw
// s_barrier, NOT __syncthreads() -- its fence would wait for vmcnt(0) and drain the prefetch (and the stores) every step
#define ZM_SYNC(N)                                                   \
  do {                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");         \
    __builtin_amdgcn_s_barrier();                                    \
    asm volatile("" ::: "memory");                                   \
  } while (0)

// WLDS = false: all 3 x KS x NT weight fragments resident in registers (fits for one input plane and one output tile);
// WLDS = true: they are copied to LDS once per workgroup (behind the ring) and every K step reads its 3 x NT fragments
// from there, double-buffered against the MFMAs -- (MT + 3 NT) LDS reads feed 3 NT MT MFMAs, < 0.6 reads per MFMA.
// STATS: per-channel sum / sum of squares of the output for the next BatchNorm (forward layers; data gradients skip the work).
// NW: waves per workgroup (4, or 8 = two per SIMD: one wave's epilogue / DMA / LDS instructions issue under its partner's MFMAs).
// ACT: 1 = bias + LeakyReLU / identity in the epilogue (forward layers), 2 = bias + ELU (the CAE's layers, Cae3D.py:41-70), 0 = the
// accumulator is stored as it is (data gradients).
// Q8: the e4m3 plane-major copy of the output is written as well (a.y8: the fp8 forward of the NEXT layer reads it, csrc/sp_conv_zm8.hip).
// HL: bf16 PAIR operands (SP_HL, the forward pass of the "bf16x3" precision mode): x = x_hi + x_lo in two tensors, the ring
// slot holds the P hi planes followed by the P lo planes, the weights come as hi and lo fragments (both in LDS), every product
// is three MFMAs (w_hi x_hi + w_hi x_lo + w_lo x_hi: ~2^-17 relative, fp32 accumulate) and the epilogue writes the output as a
// pair again.  Three times the MFMAs on twice the bytes: the 16-channel layers move from the LDS / HBM bound towards the pipe.
// STATS: 0 none; 1 per-channel sum / sum of squares of the output (forward layers); 2 (data gradients behind a BatchNorm whose
// sums cannot come from the weight gradient -- padded convolutions, the CAE): (sum g, sum g*x) of the stored output g and the
// layer input x read at the same position (a.aux), the pair the BatchNorm backward needs.  The x values of the plane whose
// epilogue rides in step i+1 are fetched into registers at the top of step i.
// STATS 3 (a.stats_mode 2; the second convolution's data gradient of a U-Net block, Unet3D.py:18-24 backward): the data gradient g
// is not stored at all -- the epilogue forms the dz of the block's FIRST convolution from it, dz = (c0 g + c1 x + c2) act'(x), x =
// a.aux = the first convolution's output (prefetched like STATS 2), with the BatchNorm-backward coefficients finalized in this
// kernel's prologue from the sums the weight gradient's finish kernel left (a.bnb; workgroup 0 adds dgamma / dbeta), and
// accumulates sum dz per channel into a.dz_sums: the separate passes sp_bn_bwd_finalize and sp_bn_act_bwd (read g, read x, write
// dz) are gone.
// BatchNorm groups (a.group_batch > 0: the batch holds B / group_batch passes with statistics of their own): the sums are flushed
// to the rows of the sample's group whenever a workgroup's march crosses into another group.
// POOL (a.pool_y; the second convolution of a down block, Unet3D.py:58-62): MaxPool3d(2) of the output rides in the epilogue --
// x pairs by DPP, y pairs from the wave's two row groups, z pairs from a register set held across one step (pieces start at even
// planes, so which step holds and which emits is known at compile time) -- and the statistics the kernel accumulates are those of the
// POOLED tensor (the next block's BatchNorm input; nobody needs the un-pooled tensor's): sp_maxpool2_fwd's pass over the largest
// activations of the network (read 8, write 1) is gone.
// SPLIT (a.y2; the data gradient of a layer whose input is a channel concatenation, Unet3D.py:66-67,71-72): output tiles [0, split_nt)
// go to y (pitch CPo), the rest to a SECOND dense tensor y2 (pitch CPo2) -- the two consumers of such a gradient (upsample backward,
// pool / skip backward) then read whole lines instead of 64 / 32 bytes of every 96-byte row, from ONE launch.
// PS ("plane-serial", a.pser_planes = PT input planes of 16 channels; round 5): layers whose weight fragments do not fit LDS beside
// a ring of whole plane SETS (48 -> 16 in the pair mode, 96 -> 32) march through (input z plane, channel plane) SUB-STEPS instead:
// the ring holds ONE 16-channel plane per slot (the kernel's P is 1), the accumulators of an output plane collect PT sub-steps per
// input plane, and the weight fragments of one channel plane (3 KS NT KiB, pairs twice that) stream through TWO LDS buffers --
// the next sub-step's fragments are DMA'd from L2 (every workgroup walks the same PT sets: they stay hot) at the top of a
// sub-step, before its share of the plane prefetch, so the counted wait of the next sub-step covers them.  Fragment order in
// memory: [plane p][(dz KS + s) NT + n] (runtime/plan.py:zm_pser_plan).
template <int P, int NT, int MT, int NSLOT, bool WLDS, int NW, int STATS, int ACT, typename TOUT, bool Q8 = false, bool HL = false, bool POOL = false, bool SPLIT = false, bool PS = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void conv_zm3_kernel(const ConvZmDev Q) {
  constexpr int WPS = NW / 4;
  constexpr int KS = (18 * P + 3) / 4;            // in-plane K steps (32 channels-taps each)
  constexpr int ITH = NW * MT + 2, ITW = 18;
  constexpr int PCH = ITH * ITW * 2;              // 16-byte chunks of one 16-channel plane
  constexpr int PP = HL ? 2 * P : P;              // bf16 planes of a ring slot
  constexpr int NCH = PP * PCH;
  constexpr int LOFF = P * PCH * 16;              // HL: byte offset of the lo planes inside a slot
  static_assert(!HL || (WLDS && ACT == 1 && !Q8 && sizeof(TOUT) == 2), "pair instances: LDS weights, bias + LeakyReLU epilogue");
  constexpr int NJ = (NCH + 64 * NW - 1) / (64 * NW);   // DMA instructions per wave and input plane
  constexpr int S = NJ * NW * 1024;               // slot stride in bytes
  constexpr int WOFF = NSLOT * S + NW * 1024;     // LDS offset of the weight fragments (WLDS), behind the ring and the dump area
  constexpr int D = NSLOT - 1;                    // prefetch distance in planes
  static_assert(!PS || (P == 1 && WLDS && !Q8 && !POOL && !SPLIT && STATS <= 1 && ACT == 1 && (NSLOT == 2 || NSLOT == 3)), "plane-serial instances: forward layers, weights streamed through LDS");
  static_assert(!SPLIT || (ACT == 0 && STATS == 0 && !Q8 && !HL && !POOL && NT >= 2 && sizeof(TOUT) == 2), "split output: plain 16-bit data gradients of two or more output tiles");
  static_assert(!POOL || (ACT == 1 && STATS == 1 && !Q8 && MT % 2 == 0 && sizeof(TOUT) == 2), "pooling epilogue: forward layers with statistics, 16-bit output, row pairs inside a wave");
  constexpr int NSP = POOL ? (MT / 2) * NT * (HL ? 2 : 1) : 0;      // pooled stores of one epilogue (issued every step: the hold steps' are dropped)
  constexpr int NS = MT * NT * ((Q8 || HL) ? 2 : 1) + NSP;      // store instructions of one epilogue
  constexpr int NA = STATS >= 2 ? MT * NT : 0;            // loads of the layer input per step (STATS 2, 3)
  constexpr int NSA = NS + NA;
  static_assert(D >= 1 && D <= 5 && (D - 1) * (NJ + NSA) <= 63, "counted vmcnt does not fit its 6-bit field");
  static_assert(STATS < 2 || (ACT == 0 && !Q8 && !HL && sizeof(TOUT) == 2 && MT * NT == 4 && D == 2), "BatchNorm-backward sums / fused dz: plain 16-bit data gradient, four tiles per wave");
  constexpr int NWF = 3 * KS * NT;                // weight fragments (1 KiB each)
  // PB (the ELU instances: the CAE's padded layers): the bias comes from a table in LDS indexed by the output voxel's border class
  // (sp_conv_args.bias_tab: BatchNorm folded per group into a padded convolution); without a table its one entry is the plain bias
  constexpr bool PB = ACT == 2;
  constexpr int WB = (HL ? 2 : 1) * NWF * 1024;      // bytes of one set of weight fragments in LDS (PS: two such buffers)
  constexpr int NJW = PS ? (((HL ? 2 : 1) * NWF + NW - 1) / NW) : 0;      // PS: weight DMA instructions per wave and sub-step
  // One wave per SIMD (NW == 4): nothing but this wave's own instruction order hides an LDS round trip.  Left alone the scheduler sinks
  // the fragment reads of K step s + 1 to just in front of their MFMAs (one register set, an lgkmcnt wait in front of every group of
  // four MFMAs: the pipe busy 43 % in 48 -> 16 @92^3); the fence keeps them above the MFMAs of step s.  With two waves per SIMD the
  // partner covers the wait and the second register set would spill.
#ifdef SP_ZM_NOFRONT
  constexpr bool ZFRONT = false;
#else
  // two ring slots: the prefetched plane's DMAs at the top of the step (they have to land within it), not between its MFMAs.  Only with
  // two waves per SIMD: a four-wave instance stalls its own MFMAs behind the burst (pair mode: +15 us per step)
  constexpr bool ZFRONT = NW == 8;
  // ... spread over the FIRST HALF of the K steps rather than in one burst (eight waves issuing six DMAs each at once: 300 cycles per
  // instruction, 24 % of the 48 -> 16 kernel with every wave stuck in the burst; tools/stamp_zm.py)
#ifdef SP_ZM_NOHALF
  constexpr bool ZHALF = false;
#else
  constexpr bool ZHALF = true;
#endif
  constexpr int KH = (KS + 1) / 2;
#ifdef SP_PS_NOHALF
  constexpr bool PSHALF = false;
#else
  constexpr bool PSHALF = true;
#endif
#endif
#ifdef SP_ZM_NOFENCE
  constexpr bool FENCE = false;
#else
  constexpr bool FENCE = NW == 4 && !PS;
#endif
#define ZM_FENCE if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
  constexpr bool PSPIPE = PS && !HL && NT == 1;      // PS: double-buffered fragment reads (the other instances spill with them)
#ifdef SP_PS_NOFRONT
  constexpr bool PSFRONT = false;
#else
  constexpr bool PSFRONT = true;      // PS with two ring slots: the prefetched plane's DMAs at the top of the sub-step, not between its MFMAs
#endif
  constexpr int BTOFF = WOFF + (WLDS ? (PS ? 2 : 1) * WB : 0);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const sp_conv_args& a = Q.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lv = lane & 15, lg = lane >> 4;
  unsigned char* ring = lds;

  // Output-channel slices in ONE launch (a.nslices > 1; see csrc/sp_conv_zm8.hip): the workgroups of an XCD split into nslices
  // teams that march over the same (column, plane) ranges at the same time -- whole output lines leave the L2, the input is
  // fetched from HBM once.
  const int nsl = a.nslices > 1 ? a.nslices : 1;
  uint32_t vb, nvb;
  int sl = 0;
  if (nsl > 1) {
    const uint32_t xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, per = (gridDim.x >> 3) / (uint32_t)nsl;      // host: gridDim.x % (8 nsl) == 0
    sl = (int)(jj % (uint32_t)nsl);
    vb = xcd * per + jj / (uint32_t)nsl;
    nvb = per * 8;
  } else {
    vb = xcd_remap(blockIdx.x, gridDim.x);
    nvb = gridDim.x;
  }
  const int c0s = sl * NT * 16;                    // first output channel of this workgroup's slice
  TOUT* const y_sl = reinterpret_cast<TOUT*>(a.y) + c0s;
  const float* const bias_sl = a.bias ? a.bias + c0s : nullptr;
  double* const stats_sl = STATS == 3 ? a.dz_sums : (a.stats ? a.stats + (size_t)c0s * 2 : nullptr);      // (STATS 3: only "is there an accumulator")

  int kv[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) kv[s] = a.ktab[s * 4 + lg];
  // this lane's voxel of column group m inside a staged plane: flattened tile index f -> (row, column); groups past the tile
  // (tw th < 16 NW MT) read voxel 0 and never store
  const int tw = Q.tw, th = Q.th, itw = Q.itw;
  int vbo[MT];
  // (row, column) of group m inside the tile, or row -1: recomputed where a piece sets up its store offsets (once per piece)
  // instead of living in 2 MT registers across the march
  auto tile_rc = [&](int m, int& fy, int& fx, bool per_piece = true) {
    int f = (wave * MT + m) * 16 + lv;
    if (per_piece) asm volatile("" : "+v"(f));      // (opaque: or the compiler hoists the results out of the piece loop -- and spills)
    fy = (int)fdiv((uint32_t)f, Q.d_tw);
    fx = f - fy * tw;
    if (f >= tw * th) { fy = -1; fx = 0; }
  };
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int fy, fx;
    tile_rc(m, fy, fx, false);
    vbo[m] = fy >= 0 ? (fy * itw + fx) * 32 : 0;
  }
  const bf16x8* __restrict__ wf = reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(a.wfrag_hi) + (size_t)sl * a.slice_wfrag_stride);
  bf16x8 w[WLDS ? 1 : 3][WLDS ? 1 : KS][WLDS ? 1 : NT];
  const unsigned char* wl = lds + WOFF + lane * 16;           // this lane's 16 bytes of fragment 0
  int w_group = 0;                                            // group whose fragments are loaded (wfrag_gstride != 0)
  if (WLDS) {
    for (int f = wave; f < NWF; f += NW) sp_dma16(reinterpret_cast<const unsigned char*>(wf) + (size_t)f * 1024 + lane * 16, lds + WOFF + f * 1024);
    if constexpr (HL) {      // the lo fragments behind the hi ones
      const unsigned char* wfl = reinterpret_cast<const unsigned char*>(a.wfrag_lo) + (size_t)sl * a.slice_wfrag_stride;
      for (int f = wave; f < NWF; f += NW) sp_dma16(wfl + (size_t)f * 1024 + lane * 16, lds + WOFF + (NWF + f) * 1024);
    }
  } else {
#pragma unroll
    for (int dz = 0; dz < (WLDS ? 0 : 3); ++dz)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) w[dz][s][n] = wf[((size_t)(dz * KS + s) * NT + n) * 64 + lane];
  }

  // per-lane DMA plan of one input plane set: chunk c = (wave + 4 j) * 64 + lane -> (plane p, row vy, voxel vx, half)
  const int xpitch = a.x_plane ? 16 : a.CPi;      // elements per voxel of one plane's row
  uint32_t rel[NJ];
  int crd[NJ];
  int lom = 0;                                    // HL: chunk j of this lane belongs to a lo plane
  const int64_t lod = HL ? a.x_lo_delta : 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + NW * j) * 64 + lane;
    const bool ok = c < NCH;
    const int cc = ok ? c : 0;
    const int pq = cc / PCH, r = cc - pq * PCH;
    const int p = (HL && pq >= P) ? pq - P : pq;
    if (HL && pq >= P) lom |= 1 << j;
    const int half = r & 1, vox = r >> 1;
    const int vy = (int)fdiv((uint32_t)vox, Q.d_itw), vx = vox - vy * itw;
    const uint32_t pl = a.x_plane ? (uint32_t)p * (uint32_t)a.x_plane : (uint32_t)p * 16u;
    rel[j] = (pl + (uint32_t)((vy * a.Wi + vx) * xpitch + half * 8)) * 2u;
    crd[j] = (vy & 0xff) | (vx << 8) | ((ok && vy < th + 2) ? 0 : (1 << 30));      // (chunks past the (th + 2) x (tw + 2) halo tile: zero page)
  }
  float bj[NT][4], s1[NT][4], s2[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) { bj[n][j] = bias_sl ? bias_sl[n * 16 + lg * 4 + j] : 0.f; s1[n][j] = s2[n][j] = 0.f; }
  const float slope = a.act == SP_ACT_NONE ? 1.f : a.act_param;       // LeakyReLU slope (identity: 1) or the ELU's alpha
  // STATS 3: the BatchNorm-backward coefficients of this lane's channels.  Channel c = tid / 16 is finalized by 16 lanes that split
  // the replica rows (bn_bwd_finalize_kernel's arithmetic), handed over through LDS (the ring is not in use yet).
  float cf0[STATS == 3 ? NT : 1][4], cf1[STATS == 3 ? NT : 1][4], cf2[STATS == 3 ? NT : 1][4];
  if constexpr (STATS == 3) {
    static_assert(STATS != 3 || 64 * NW / 16 >= NT * 16, "one 16-lane group per output channel");
    const sp_bn_bwd_args& bb = a.bnb;
    float* cfl = reinterpret_cast<float*>(lds);      // [3][NT * 16]
    const int c = tid >> 4, sub = tid & 15;
    if (c < NT * 16) {
      double t1 = 0, t2 = 0;
      if (c < bb.C)
        for (int r = sub; r < bb.nrep; r += 16) { t1 += bb.sums[((size_t)r * bb.CP + c) * 2]; t2 += bb.sums[((size_t)r * bb.CP + c) * 2 + 1]; }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 16); t2 += __shfl_xor(t2, o, 16); }
      if (sub == 0) {
        float k0 = 0.f, k1 = 0.f, k2 = 0.f;
        if (c < bb.C) {
          const double mu = bb.mean[c], is = bb.invstd[c], ga = bb.gamma[c];
          const double dg = (t2 - mu * t1) * is, db = t1;
          const double c0 = ga * is, c1 = -ga * is * is * dg / bb.count;
          k0 = (float)c0; k1 = (float)c1; k2 = (float)(-c0 * db / bb.count - c1 * mu);
          if (blockIdx.x == 0 && bb.dgamma) { bb.dgamma[c] += bb.pscale * (float)dg; bb.dbeta[c] += bb.pscale * (float)db; }
        }
        cfl[c] = k0; cfl[NT * 16 + c] = k1; cfl[2 * NT * 16 + c] = k2;
        if (blockIdx.x == 0 && bb.coef && c < bb.CP) { bb.coef[c] = k0; bb.coef[bb.CP + c] = k1; bb.coef[2 * bb.CP + c] = k2; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cf0[n][j] = cfl[n * 16 + lg * 4 + j]; cf1[n][j] = cfl[NT * 16 + n * 16 + lg * 4 + j]; cf2[n][j] = cfl[2 * NT * 16 + n * 16 + lg * 4 + j];
      }
    __syncthreads();      // (every lane holds its coefficients before the first plane lands in the ring)
  }
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(Q.zeros);
  // statistics flush: wave-ordered sum through LDS, one atomic per channel and workgroup into the rows of BatchNorm group g
  const int gbatch = a.group_batch > 0 ? a.group_batch : 0x7fffffff;
  int cur_g = -1;
  auto flush_stats = [&](int g) {
    float* red = reinterpret_cast<float*>(lds);      // [NW waves][NT * 32] (ordered sum: sp_cols_sum)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x1s = row16_sum(s1[n][j]), x2s = row16_sum(s2[n][j]);
        if (lv == 0) { red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2] = x1s; red[wave * (NT * 32) + (n * 16 + lg * 4 + j) * 2 + 1] = x2s; }
        s1[n][j] = s2[n][j] = 0.f;
      }
    __syncthreads();
    if constexpr (STATS == 3) {      // sum dz into the replica rows of the first convolution's bias-gradient accumulator
      for (int k = tid; k < NT * 32; k += 64 * NW) {
        const int c = k >> 1;
        if ((k & 1) == 0 && c0s + c < a.CPo) atomicAdd(&a.dz_sums[(size_t)(blockIdx.x & (SP_REDUCE_ROWS - 1)) * a.CPo + c0s + c], (double)sp_cols_sum(red, NT * 32, NW, k));
      }
    } else {
      double* const gs = stats_sl + (size_t)g * a.stats_nrep * a.CPo * 2;
      for (int k = tid; k < NT * 32; k += 64 * NW) {
        const int c = k >> 1;
        if (c0s + c < a.CPo) atomicAdd(&gs[(size_t)(blockIdx.x & (a.stats_nrep - 1)) * a.CPo * 2 + (size_t)c * 2 + (k & 1)], (double)sp_cols_sum(red, NT * 32, NW, k));
      }
    }
  };
  // PB: border classes per axis (2 pad + 1; forward convolutions: o0 = -pad), 1 x 1 x 1 without a table
  const int bt_pz = (PB && a.bias_tab) ? -a.o0D : 0, bt_py = (PB && a.bias_tab) ? -a.o0H : 0, bt_px = (PB && a.bias_tab) ? -a.o0W : 0;
  const int bt_ny = 2 * bt_py + 1, bt_nx = 2 * bt_px + 1, bt_ncls = (2 * bt_pz + 1) * bt_ny * bt_nx;
  auto bt_class = [](int o, int p, int n) { return o < p ? o : (o >= n - p ? min(2 * p, p + 1 + o - (n - p)) : p); };
  zm_u32x2 ax[2][4];                                // STATS 2: x at this lane's four (row, tile) positions, two planes in flight
  uint32_t axtok = 0;                               // (see the wait in ZM_STEP)
#ifdef SP_ZM_STAMPS
  unsigned long long zm_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // 0 sync, 1 DMA issue, 2 K loop + epilogue, 3 steps, 4 total
  ZM_T(t_begin);
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif

  // Work = (column, output plane) pairs cut into gridDim.x equal pieces of the flattened sequence; a piece that starts
  // inside a column pays the two-plane prologue again.  XCD-aware piece id (neighbouring columns share halo in one L2).
  // (POOL: the unit is a PAIR of planes, so every piece starts at an even plane)
  const uint32_t DoU = POOL ? (uint32_t)(a.Do + 1) / 2 : (uint32_t)a.Do;
  const uint64_t T = (uint64_t)Q.ncols * DoU;
  uint64_t pos = T * vb / nvb;
  const uint64_t pend_pos = T * (vb + 1) / nvb;
  float phold[POOL ? MT / 2 : 1][POOL ? NT : 1][4];      // POOL: (x, y)-pooled values of the even plane, until the odd one's epilogue
  while (pos < pend_pos) {
    const uint32_t col = (uint32_t)(pos / DoU);
    const int u0 = (int)(pos - (uint64_t)col * DoU);
    const int u1 = (int)min((uint64_t)DoU, (uint64_t)u0 + (pend_pos - pos));
    pos += (uint64_t)(u1 - u0);
    const int z0 = POOL ? 2 * u0 : u0, z1 = POOL ? min(a.Do, 2 * u1) : u1;
    const int nz = z1 - z0, nin = nz + 2;                     // output planes of this piece, input planes they need
    uint32_t t = col;
    uint32_t q = fdiv(t, Q.d_tx); const int tx = t - q * Q.ntx; t = q;
    q = fdiv(t, Q.d_ty); const int ty = t - q * Q.nty; const int b = q;
    {
      const int g = b / gbatch;
      if (g != cur_g) {
        if (STATS && stats_sl != nullptr && cur_g >= 0) {       // the march enters another BatchNorm group: hand over what belongs to the last one
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          flush_stats(cur_g);
        }
        if constexpr (PB) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();                                     // every wave is past its last epilogue / K step of the old group
          if (a.wfrag_gstride != 0 && g != w_group) {          // the group's own folded weights
            const bf16x8* wg = reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(wf) + (size_t)g * a.wfrag_gstride);
            if (WLDS) {
              for (int f = wave; f < NWF; f += NW) sp_dma16(reinterpret_cast<const unsigned char*>(wg) + (size_t)f * 1024 + lane * 16, lds + WOFF + f * 1024);
            } else {
#pragma unroll
              for (int dz = 0; dz < (WLDS ? 0 : 3); ++dz)
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                  for (int n = 0; n < NT; ++n) w[dz][s][n] = wg[((size_t)(dz * KS + s) * NT + n) * 64 + lane];
            }
            w_group = g;
          }
          // bias table of the group: [class][NT * 16] floats (one class: the plain bias)
          float* bt = reinterpret_cast<float*>(lds + BTOFF);
          const int nent = bt_ncls * NT * 16;
          for (int k = tid; k < nent; k += 64 * NW) {
            const int c = k % (NT * 16), cl = k / (NT * 16);
            bt[k] = a.bias_tab ? a.bias_tab[(size_t)g * a.bias_tab_gstride + (size_t)cl * a.CPo + c0s + c] : (bias_sl ? bias_sl[c] : 0.f);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        } else if (STATS && stats_sl != nullptr && cur_g >= 0) {
          __syncthreads();
        }
        cur_g = g;
      }
    }
    const int oy0 = ty * th, ox0 = tx * tw;
    const int iy0 = oy0 + a.o0H, ix0 = ox0 + a.o0W;
    const unsigned char* auxb = STATS >= 2 ? reinterpret_cast<const unsigned char*>(a.aux) + (size_t)b * a.YD * a.YH * a.YW * a.CPo * 2 : nullptr;
    const unsigned char* xin = reinterpret_cast<const unsigned char*>(a.x) + (size_t)b * a.Di * a.Hi * a.Wi * xpitch * 2;
    int vmask = 0;                                            // in-plane validity of this lane's chunks
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int vy = crd[j] & 0xff, vx = (crd[j] >> 8) & 0xff;
      if (!(crd[j] >> 30) && (unsigned)(iy0 + vy) < (unsigned)a.Hi && (unsigned)(ix0 + vx) < (unsigned)a.Wi) vmask |= 1 << j;
    }
    // one DMA instruction (1 KiB) of input plane i -> ring slot; inloop: issued between the MFMAs of a K step
    const unsigned char* pl_src0 = nullptr;
    unsigned char* pl_dst0 = nullptr;
    int pl_mask = 0;
    bool pl_fill = false;                                     // filler DMAs (zero page -> dump area) beyond the last plane
    const int64_t ps_plane = a.x_plane ? a.x_plane * 2 : 32;      // PS: bytes from channel plane p to p + 1 (plane-major / channels-last)
    auto plane_begin = [&](int i, int slot, int pch = 0) {
      const int iz = z0 + a.o0D + i;
      pl_mask = ((unsigned)iz < (unsigned)a.Di) ? vmask : 0;
      pl_src0 = xin + (PS ? (int64_t)pch * ps_plane : (int64_t)0) + (((int64_t)iz * a.Hi + iy0) * a.Wi + ix0) * (int64_t)(xpitch * 2);
      pl_dst0 = ring + slot * S + wave * 1024;
    };
    // PS: the fragments of channel plane pch into weight buffer `buf` (pch < 0: fillers only): NJW instructions in every wave
    const unsigned char* const wfb = reinterpret_cast<const unsigned char*>(a.wfrag_hi);
    const unsigned char* const wflb = reinterpret_cast<const unsigned char*>(a.wfrag_lo);
    auto w_dma = [&](int pch, int buf, bool inloop) {
#pragma unroll
      for (int k = 0; k < NJW; ++k) {
        const int f = wave + NW * k;
        const bool ok = pch >= 0 && f < (HL ? 2 : 1) * NWF;
        const unsigned char* src = !ok ? zsrc : ((HL && f >= NWF) ? wflb + ((size_t)pch * NWF + (f - NWF)) * 1024 + lane * 16 : wfb + ((size_t)pch * NWF + f) * 1024 + lane * 16);
        unsigned char* dst = ok ? lds + WOFF + buf * WB + f * 1024 : ring + NSLOT * S + wave * 1024;
        if (inloop) sp_dma16_nc(src, dst); else sp_dma16(src, dst);
      }
    };
    auto plane_dma = [&](int j, bool inloop) {
      const unsigned char* src = ((pl_mask >> j) & 1) ? pl_src0 + rel[j] + (HL && ((lom >> j) & 1) ? lod : (int64_t)0) : zsrc;      // padding / overhang: the zero page
      unsigned char* dst = pl_dst0 + (pl_fill ? 0 : j * (NW * 1024));
      if (inloop) sp_dma16_nc(src, dst); else sp_dma16(src, dst);
    };
    auto load_plane = [&](int i, int slot) {                  // whole plane at once (prologue)
      plane_begin(i, slot);
#pragma unroll
      for (int j = 0; j < NJ; ++j) plane_dma(j, false);
    };
    TOUT* yout = y_sl + (size_t)b * a.YD * a.YH * a.YW * a.CPo;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)yout, 0, (int)((uint32_t)a.YD * a.YH * a.YW * a.CPo * (uint32_t)sizeof(TOUT)), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs_lo = __builtin_amdgcn_make_buffer_rsrc(      // HL: the lo halves of the output
        (void*)(reinterpret_cast<unsigned char*>(yout) + (HL ? a.y_lo_delta : 0)), 0,
        (int)((uint32_t)a.YD * a.YH * a.YW * a.CPo * (uint32_t)sizeof(TOUT)), 0x00020000);
    // byte offset of (column group m: this lane's voxel and channel quad) inside an output plane, or "outside"
    uint32_t rowoff[MT];
    uint32_t rowok[MT];       // all-ones / zero bit mask (a float 0/1 factor would turn the garbage of a row outside the output into NaN)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      int fy, fx;
      tile_rc(m, fy, fx);
      const int oy = oy0 + fy, ox = ox0 + fx;
      const bool ok = fy >= 0 && ox < a.Wo && oy < a.Ho;
      rowoff[m] = ok ? (uint32_t)((((oy * a.osH + a.ooH) * a.YW + (ox * a.osW + a.ooW)) * a.CPo + lg * 4) * (int)sizeof(TOUT)) : 0x80000000u;
      rowok[m] = ok ? 0xffffffffu : 0u;
    }
    // SPLIT: the second output tensor of sample b and this lane's offsets in it (same voxels, its own channel pitch)
    const __amdgpu_buffer_rsrc_t yrs2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(SPLIT ? reinterpret_cast<unsigned char*>(a.y2) + (size_t)b * a.YD * a.YH * a.YW * a.CPo2 * sizeof(TOUT) : nullptr), 0,
        (int)(SPLIT ? (uint32_t)a.YD * a.YH * a.YW * a.CPo2 * (uint32_t)sizeof(TOUT) : 0u), 0x00020000);
    uint32_t rowoff2[SPLIT ? MT : 1];
    if constexpr (SPLIT) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int fy, fx;
        tile_rc(m, fy, fx);
        const int oy = oy0 + fy, ox = ox0 + fx;
        rowoff2[m] = (fy >= 0 && ox < a.Wo && oy < a.Ho) ? (uint32_t)(((oy * a.YW + ox) * a.CPo2 + lg * 4) * (int)sizeof(TOUT)) : 0x80000000u;
      }
    }
    const uint32_t zstride2 = (uint32_t)(a.YH * a.YW * a.CPo2 * (int)sizeof(TOUT));
    int btrow[PB ? MT : 1];                          // PB: byte offset of (row class, column class, this lane's channel quad) in the bias table
    if constexpr (PB) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int fy, fx;
        tile_rc(m, fy, fx);
        const int oy = min(oy0 + max(fy, 0), a.Ho - 1), ox = min(ox0 + fx, a.Wo - 1);
        btrow[m] = ((bt_class(oy, bt_py, a.Ho) * bt_nx + bt_class(ox, bt_px, a.Wo)) * (NT * 16) + lg * 4) * 4;
      }
    }
    // POOL: the pooled tensor of sample b [YD/2][YH/2][YW/2][CPo] (HL: its lo half pool_lo_delta bytes behind) and the offset of
    // this lane's pooled voxel per row pair (even lanes of even rows whose 2 x 2 window lies inside the output)
    const int Dp = a.YD >> 1, Hp = a.YH >> 1, Wp = a.YW >> 1;
    const uint32_t psz = (uint32_t)Dp * Hp * Wp * a.CPo * 2u;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(POOL ? reinterpret_cast<unsigned char*>(a.pool_y) + (size_t)b * psz : nullptr), 0, (int)(POOL ? psz : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t prs_lo = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((POOL && HL) ? reinterpret_cast<unsigned char*>(a.pool_y) + a.pool_lo_delta + (size_t)b * psz : nullptr), 0, (int)((POOL && HL) ? psz : 0u), 0x00020000);
    uint32_t poff[POOL ? MT / 2 : 1];
    if constexpr (POOL) {
#pragma unroll
      for (int mp = 0; mp < MT / 2; ++mp) {
        const int oy = oy0 + wave * MT + 2 * mp, ox = ox0 + lv;      // (classic tile: checked by the host)
        const bool ok = (lv & 1) == 0 && (oy >> 1) < Hp && (ox >> 1) < Wp;
        poff[mp] = ok ? (uint32_t)((((oy >> 1) * Wp + (ox >> 1)) * a.CPo + lg * 4) * 2) : 0x80000000u;
      }
    }
    const uint32_t pzstride = (uint32_t)(Hp * Wp * a.CPo * 2);
    const uint32_t zstride = (uint32_t)(a.osD * a.YH * a.YW * a.CPo * (int)sizeof(TOUT));
    const uint32_t zbase = (uint32_t)(a.ooD * a.YH * a.YW * a.CPo * (int)sizeof(TOUT));
    // the e4m3 copy (dense output only: no phase strides): plane n of sample b, 4 bytes per lane at voxel * 16 + lg * 4
    __amdgpu_buffer_rsrc_t y8rs[Q8 ? NT : 1];
    uint32_t rowoff8[Q8 ? MT : 1];
    const uint32_t zstride8 = (uint32_t)(a.YH * a.YW * 16);
    const float q8s = a.y8_scale;
    if constexpr (Q8) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        y8rs[n] = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<unsigned char*>(a.y8) + (size_t)n * a.y8_plane + (size_t)b * a.YD * a.YH * a.YW * 16), 0,
            (int)((uint32_t)a.YD * a.YH * a.YW * 16u), 0x00020000);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int fy, fx;
        tile_rc(m, fy, fx);
        const int oy = oy0 + fy, ox = ox0 + fx;
        rowoff8[m] = (fy >= 0 && ox < a.Wo && oy < a.Ho) ? (uint32_t)((oy * a.YW + ox) * 16 + lg * 4) : 0x80000000u;
      }
    }

    // FOUR accumulator sets, one per output plane j mod 4: during step i (input plane i) the sets of planes i, i-1, i-2
    // receive the taps dz = 0, 1, 2 while the set of plane i-3 -- finished at the end of step i-1 -- is read by the epilogue
    // that is carried in this step's instruction stream.  No copy into a staging set, and no zeroing either: the first MFMA
    // of a plane (dz = 0, K step 0) takes a zero C operand.
    f32x4 acc[4][NT][MT];

    const int PT = PS ? a.pser_planes : 1;                    // channel planes per input z plane (PS)
    ZM_SYNC(0);                                               // the previous piece has been consumed
    if constexpr (PS) {
#pragma unroll
      for (int k = 0; k < NSLOT - 1; ++k) {                   // sub-steps 0 .. D-1: (input plane k / PT, channel plane k % PT)
        const int ik = k / PT, pk = k - ik * PT;
        if (ik < nin) {
          plane_begin(ik, k, pk);
#pragma unroll
          for (int j = 0; j < NJ; ++j) plane_dma(j, false);
        }
      }
      w_dma(0, 0, false);
    } else {
#pragma unroll
      for (int k = 0; k < NSLOT - 1; ++k)
        if (k < nin) load_plane(k, k);
    }

    // epilogue of the plane held by accumulator set R: straight-line code (no branch); fz < 0 (no finished plane) turns
    // every store into an out-of-range one and every statistics term into 0 -- the instruction count never changes
#define ZM_EPILOGUE(R_, fz_, AX_, EMIT_)                                                                            \
  {                                                                                                               \
    const bool pv = (fz_) >= 0;                                                                                   \
    /* POOL: plane fz is odd in the EMIT_ steps (pieces start at even planes): its pooled plane (fz_ >> 1), if inside */ \
    const uint32_t pzo_ = (POOL && (EMIT_) && pv && ((fz_) >> 1) < Dp) ? (uint32_t)((fz_) >> 1) * pzstride : 0x80000000u; \
    const uint32_t zoff = pv ? zbase + (uint32_t)(fz_) * zstride : 0x80000000u;                                   \
    const uint32_t pm = pv ? 0xffffffffu : 0u;                                                                    \
    const unsigned char* btz_ = lds + BTOFF + (PB ? bt_class(pv ? (fz_) : 0, bt_pz, a.Do) * (bt_ny * bt_nx * NT * 64) : 0); \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                              \
      const uint32_t off = (zoff | rowoff[m]) & 0x80000000u ? 0x80000000u : zoff + rowoff[m];                     \
      const uint32_t off8 = !Q8 || ((zoff | rowoff[m]) & 0x80000000u) ? 0x80000000u : (uint32_t)(fz_) * zstride8 + rowoff8[Q8 ? m : 0]; \
      const uint32_t msk = pm & rowok[m];                                                                         \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                            \
        float v[4];                                                                                               \
        f32x4 bq_ = {0.f, 0.f, 0.f, 0.f};                                                                         \
        if constexpr (PB) bq_ = *reinterpret_cast<const f32x4*>(btz_ + btrow[PB ? m : 0] + n * 64);               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
          if (ACT == 1) { const float zz = acc[R_][n][m][j] + bj[n][j]; v[j] = fmaxf(zz, slope * zz); }           \
          else if (ACT == 2) { const float zz = acc[R_][n][m][j] + bq_[j]; v[j] = zz > 0.f ? zz : slope * (__expf(zz) - 1.f); } \
          else v[j] = acc[R_][n][m][j];                                                                           \
        }                                                                                                         \
        if constexpr (STATS == 3) {   /* dz = (c0 g + c1 x + c2) act'(x) of the ROUNDED g (what sp_bn_act_bwd would read back) */ \
          const uint32_t g0_ = zm_pack2(v[0], v[1]), g1_ = zm_pack2(v[2], v[3]);                                  \
          const float gr_[4] = {sp_h2f_lo(g0_), sp_h2f_hi(g0_), sp_h2f_lo(g1_), sp_h2f_hi(g1_)};                  \
          const zm_u32x2 xw_ = {ax[AX_][n * MT + m][0] | axtok, ax[AX_][n * MT + m][1] | axtok};                     \
          const float xq_[4] = {sp_h2f_lo(xw_[0]), sp_h2f_hi(xw_[0]), sp_h2f_lo(xw_[1]), sp_h2f_hi(xw_[1])};      \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const float o_ = (cf0[STATS == 3 ? n : 0][j] * gr_[j] + cf1[STATS == 3 ? n : 0][j] * xq_[j] + cf2[STATS == 3 ? n : 0][j]) * (xq_[j] > 0.f ? 1.f : slope); \
            v[j] = o_;                                                                                            \
            s1[n][j] += __uint_as_float(__float_as_uint(o_) & msk);                                               \
          }                                                                                                       \
        }                                                                                                         \
        if constexpr (HL) {                                                                                       \
          uint32_t h0_, h1_, l0_, l1_;                                                                            \
          sp_hl_split4(v, h0_, h1_, l0_, l1_);                                                                    \
          const zm_u32x2 dh_ = {h0_, h1_}, dl_ = {l0_, l1_};                                                      \
          __builtin_amdgcn_raw_buffer_store_b64(dh_, yrs, off + (uint32_t)(n * 32), 0, 0);                        \
          __builtin_amdgcn_raw_buffer_store_b64(dl_, yrs_lo, off + (uint32_t)(n * 32), 0, 0);                     \
        } else if constexpr (SPLIT) {      /* tiles past split_nt: the second tensor (uniform select, no branch) */                \
          const bool sec_ = n >= a.split_nt;                                                                      \
          const uint32_t o2_ = (!pv || (rowoff2[SPLIT ? m : 0] & 0x80000000u)) ? 0x80000000u : (uint32_t)(fz_) * zstride2 + rowoff2[SPLIT ? m : 0]; \
          const uint32_t os_ = sec_ ? ((o2_ & 0x80000000u) ? o2_ : o2_ + (uint32_t)((n - a.split_nt) * 16 * (int)sizeof(TOUT))) : off + (uint32_t)(n * 16 * (int)sizeof(TOUT)); \
          ZmStore<TOUT>::st4(sec_ ? yrs2 : yrs, os_, v);                                                          \
        } else ZmStore<TOUT>::st4(yrs, off + (uint32_t)(n * 16 * (int)sizeof(TOUT)), v);                          \
        if constexpr (Q8) {      /* of the STORED 16-bit values: the copy equals sp_quantize_f8 of y bit for bit */     \
          const uint32_t w0_ = zm_pack2(v[0], v[1]), w1_ = zm_pack2(v[2], v[3]);                                  \
          const float r_[4] = {sp_h2f_lo(w0_), sp_h2f_hi(w0_), sp_h2f_lo(w1_), sp_h2f_hi(w1_)};                   \
          __builtin_amdgcn_raw_buffer_store_b32(zm_pack4_e4m3(r_, q8s), y8rs[n], off8, 0, 0);                     \
        }                                                                                                         \
        if constexpr (STATS == 1 && !POOL) {   /* of the fp32 values (what an fp32 BatchNorm would see; the bf16 rounding averages out) */  \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const float u = __uint_as_float(__float_as_uint(v[j]) & msk);                                         \
            s1[n][j] += u; s2[n][j] = fmaf(u, u, s2[n][j]);                                                       \
          }                                                                                                       \
        }                                                                                                         \
        if constexpr (STATS == 2) {   /* (sum g, sum g x) of the STORED g, as the tiled kernel's stats_mode 1 */        \
          const uint32_t w0_ = zm_pack2(v[0], v[1]), w1_ = zm_pack2(v[2], v[3]);                                  \
          const float r_[4] = {sp_h2f_lo(w0_), sp_h2f_hi(w0_), sp_h2f_lo(w1_), sp_h2f_hi(w1_)};                   \
          const zm_u32x2 xw_ = {ax[AX_][n * MT + m][0] | axtok, ax[AX_][n * MT + m][1] | axtok};                     \
          const float xq_[4] = {sp_h2f_lo(xw_[0]), sp_h2f_hi(xw_[0]), sp_h2f_lo(xw_[1]), sp_h2f_hi(xw_[1])};      \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const float u = __uint_as_float(__float_as_uint(r_[j]) & msk);                                        \
            const float xx = __uint_as_float(__float_as_uint(xq_[j]) & msk);                                      \
            s1[n][j] += u; s2[n][j] = fmaf(u, xx, s2[n][j]);                                                      \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    /* POOL: a loop nest of its own (the activation of the two row groups is recomputed from the accumulators: a few VALU */ \
    /* instructions instead of 8 NT live registers across the row loop above) */                                   \
    if constexpr (POOL) {                                                                                         \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                              \
        _Pragma("unroll") for (int mp = 0; mp < MT / 2; ++mp) {                                                   \
          float o_[4];                                                                                            \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const float za_ = acc[R_][n][2 * mp][j] + bj[n][j], zb_ = acc[R_][n][2 * mp + 1][j] + bj[n][j];       \
            float pm_ = fmaxf(fmaxf(za_, slope * za_), fmaxf(zb_, slope * zb_));      /* y pair: row groups 2 mp, 2 mp + 1 */ \
            pm_ = fmaxf(pm_, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, pm_), 0xB1, 0xF, 0xF, true)));   /* x pair: lanes lv, lv ^ 1 */ \
            o_[j] = (EMIT_) ? fmaxf(phold[POOL ? mp : 0][POOL ? n : 0][j], pm_) : pm_;      /* z pair: the plane held since the last step */ \
            if (!(EMIT_)) phold[POOL ? mp : 0][POOL ? n : 0][j] = pm_;                                             \
          }                                                                                                       \
          const uint32_t po_ = ((pzo_ | poff[POOL ? mp : 0]) & 0x80000000u) ? 0x80000000u : pzo_ + poff[POOL ? mp : 0] + (uint32_t)(n * 32); \
          const uint32_t pmsk_ = (po_ & 0x80000000u) ? 0u : 0xffffffffu;                                          \
          if constexpr (HL) {                                                                                     \
            uint32_t h0_, h1_, l0_, l1_;                                                                          \
            sp_hl_split4(o_, h0_, h1_, l0_, l1_);                                                                 \
            const zm_u32x2 dh_ = {h0_, h1_}, dl_ = {l0_, l1_};                                                    \
            __builtin_amdgcn_raw_buffer_store_b64(dh_, prs, po_, 0, 0);                                           \
            __builtin_amdgcn_raw_buffer_store_b64(dl_, prs_lo, po_, 0, 0);                                        \
            if (EMIT_) {      /* (compile time: the hold steps' masked zero terms are not folded away -- and spill) */    \
              _Pragma("unroll") for (int j = 0; j < 4; ++j) {      /* statistics of the pair values */             \
                const float u = __uint_as_float(__float_as_uint(o_[j]) & pmsk_);                                  \
                s1[n][j] += u; s2[n][j] = fmaf(u, u, s2[n][j]);                                                   \
              }                                                                                                   \
            }                                                                                                     \
          } else {                                                                                                \
            const zm_u32x2 d_ = {zm_pack2(o_[0], o_[1]), zm_pack2(o_[2], o_[3])};                                 \
            __builtin_amdgcn_raw_buffer_store_b64(d_, prs, po_, 0, 0);                                            \
            if (EMIT_) {                                                                                          \
              const float r_[4] = {sp_h2f_lo(d_[0]), sp_h2f_hi(d_[0]), sp_h2f_lo(d_[1]), sp_h2f_hi(d_[1])};      /* of the STORED values, as sp_maxpool2_fwd */ \
              _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
                const float u = __uint_as_float(__float_as_uint(r_[j]) & pmsk_);                                  \
                s1[n][j] += u; s2[n][j] = fmaf(u, u, s2[n][j]);                                                   \
              }                                                                                                   \
            }                                                                                                     \
          }                                                                                                       \
        }                                                                                                         \
    }                                                                                                             \
  }

    // the activation fragment of column group m: slot + this lane's voxel of the group + the step's octet
#define ZM_LDX(dst, dstl, s_)                                                                                     \
  {                                                                                                               \
    const unsigned char* xa_ = sb + kv[s_];                                                                       \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(xa_ + vbo[m]);        \
    if constexpr (HL) {                                                                                           \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) dstl[m] = *reinterpret_cast<const bf16x8*>(xa_ + LOFF + vbo[m]); \
    }                                                                                                             \
  }
#define ZM_LDW(dst, dstl, s_)                                                                                     \
  if (WLDS) {                                                                                                     \
    _Pragma("unroll") for (int dz = 0; dz < 3; ++dz)                                                              \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                          \
            dst[dz][n] = *reinterpret_cast<const bf16x8*>(wl + ((dz * KS + (s_)) * NT + n) * 1024);               \
            if constexpr (HL) dstl[dz][n] = *reinterpret_cast<const bf16x8*>(wl + (NWF + (dz * KS + (s_)) * NT + n) * 1024); \
        }                                                                                                         \
  }
#define ZM_LDW_B(dst, dstl, s_, wb_)                                                                              \
  {                                                                                                               \
    _Pragma("unroll") for (int dz = 0; dz < 3; ++dz)                                                              \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                          \
            dst[dz][n] = *reinterpret_cast<const bf16x8*>((wb_) + ((dz * KS + (s_)) * NT + n) * 1024);            \
            if constexpr (HL) dstl[dz][n] = *reinterpret_cast<const bf16x8*>((wb_) + (NWF + (dz * KS + (s_)) * NT + n) * 1024); \
        }                                                                                                         \
  }
#define ZM_DMA_(s_) _Pragma("unroll") for (int j = ((s_) * NJ) / KS; j < (((s_) + 1) * NJ) / KS; ++j) plane_dma(j, true);
#define ZM_DMA_H(s_) _Pragma("unroll") for (int j = ((s_) * NJ) / KH; j < ((s_) < KH ? (((s_) + 1) * NJ) / KH : 0); ++j) plane_dma(j, true);
#define ZM_DMA(s_) if constexpr (D == 1 && ZFRONT && !PS) { if constexpr (ZHALF) { ZM_DMA_H(s_) } } else { ZM_DMA_(s_) }
#define ZM_W(DZ_, s_, n_, wv) (WLDS ? wv[DZ_][n_] : w[WLDS ? 0 : DZ_][WLDS ? 0 : s_][WLDS ? 0 : n_])
    // HL: the two cross terms follow the hi x hi product into the same accumulator
#define ZM_MMA_X(R_, DZ_, xv, xvl, wv, wvl)                                                                       \
          if constexpr (HL) {                                                                                     \
            acc[R_][n][m] = SP_MFMA16(wv[DZ_][n], xvl[m], acc[R_][n][m], 0, 0, 0);                                \
            acc[R_][n][m] = SP_MFMA16(wvl[DZ_][n], xv[m], acc[R_][n][m], 0, 0, 0);                                \
          }
#define ZM_MMA(R_, DZ_, s_, xv, xvl, wv, wvl)                                                                     \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                  \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                            \
          acc[R_][n][m] = SP_MFMA16(ZM_W(DZ_, s_, n, wv), xv[m], acc[R_][n][m], 0, 0, 0);                         \
          ZM_MMA_X(R_, DZ_, xv, xvl, wv, wvl)                                                                     \
      }
    // first contribution to a new output plane: C = 0
#define ZM_MMA0(R_, DZ_, s_, xv, xvl, wv, wvl)                                                                    \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                  \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                            \
          acc[R_][n][m] = SP_MFMA16(ZM_W(DZ_, s_, n, wv), xv[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);             \
          ZM_MMA_X(R_, DZ_, xv, xvl, wv, wvl)                                                                     \
      }
#define ZM_MMA_D0(R_, s_, xv, xvl, wv, wvl) if ((s_) == 0) { ZM_MMA0(R_, 0, s_, xv, xvl, wv, wvl) } else { ZM_MMA(R_, 0, s_, xv, xvl, wv, wvl) }
    // One step = one input plane i (phase PH = i mod 4, compile time): taps dz = 0 / 1 / 2 add into the sets PH, PH+3, PH+2
    // (mod 4) of the output planes i, i-1, i-2; set PH+1 holds plane i-3, whose epilogue is issued between this step's MFMAs.
#define ZM_STEP(PH)                                                                                               \
  {                                                                                                               \
    /* plane i has landed (issued NSLOT-1 steps ago).  Every step issues exactly NJ DMA instructions (of a later plane) and */ \
    /* MT*NT stores (out-of-range ones are dropped by the buffer unit but counted), interleaved with its MFMAs in an order */ \
    /* the compiler picks: everything issued in the last D-1 steps may stay in flight.  The barrier makes "landed" true for */ \
    /* every wave, and everyone is done with the slot that is refilled during this step. */                         \
    /* younger than plane i's DMA: the DMAs of planes i+1 .. i+D-1 and the stores of the last min(i, D-1) steps */   \
    ZM_T(ts0);                                                                                                    \
    if (i >= D - 1) ZM_SYNC((D - 1) * (NJ + NSA));                                                                \
    else if (i == 0) ZM_SYNC((D - 1) * NJ);                                                                       \
    else if (i == 1) ZM_SYNC((D - 1) * NJ + (D > 2 ? 1 : 0) * NSA);                                               \
    else if (i == 2) ZM_SYNC((D - 1) * NJ + (D > 3 ? 2 : 0) * NSA);                                               \
    else ZM_SYNC((D - 1) * NJ + (D > 4 ? 3 : 0) * NSA);                                                           \
    if constexpr (STATS >= 2) {                                                                                   \
      /* the x values for THIS step's epilogue were requested at the top of the last step, before its NJ DMAs and NS stores: */ \
      /* all but those may be outstanding.  The wait also defines a zero token that every use of the loaded registers ORs in, */ \
      /* so that none of them can be scheduled above it ("+v" ties on the registers themselves made the allocator spill). */   \
      asm volatile("s_waitcnt vmcnt(%1)\n\tv_mov_b32 %0, 0" : "=v"(axtok) : "n"(NJ + NS) : "memory");               \
      /* ... and the ones for the next step's epilogue (output plane z0 + i - 2) go out now; outside the output: the zero page */ \
      const int fzn = (i >= 2 && i - 2 < nz) ? z0 + i - 2 : -1;                                                   \
      const uint32_t zoffn = fzn >= 0 ? zbase + (uint32_t)fzn * zstride : 0x80000000u;                            \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                              \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                          \
          const unsigned char* ap_ = ((zoffn | rowoff[m]) & 0x80000000u) ? zsrc : auxb + (zoffn + rowoff[m] + (uint32_t)(n * 32)); \
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ax[((PH) + 1) & 1][n * MT + m]) : "v"(ap_) : "memory"); \
        }                                                                                                         \
    }                                                                                                             \
    ZM_T(ts1);                                                                                                    \
    /* the DMA of plane i+D goes out piecewise between the MFMAs below (ZM_DMA); beyond the piece's last plane the same */ \
    /* number of instructions copies the zero page into a dump area, so that the counted waits stay valid */          \
    pl_fill = false;                                                                                              \
    if (i + D < nin) plane_begin(i + D, (islot + D) % NSLOT);                                                     \
    else { pl_mask = 0; pl_fill = true; pl_dst0 = ring + NSLOT * S + wave * 1024; }                               \
    /* two ring slots (prefetch distance 1): the whole plane goes out NOW -- it has to land within this step */     \
    if constexpr (D == 1 && ZFRONT && !ZHALF) { _Pragma("unroll") for (int j = 0; j < NJ; ++j) plane_dma(j, true); } \
    ZM_T(ts2);                                                                                                    \
    const unsigned char* sb = ring + islot * S;                                                                   \
    const bool v0 = i < nz, v1 = i >= 1 && i - 1 < nz, v2 = i >= 2 && i - 2 < nz;                                 \
    const int fz = (i >= 3 && i - 3 < nz) ? z0 + i - 3 : -1;       /* the plane whose epilogue rides in this step */ \
    bf16x8 x0[MT], x1[MT], x0l[HL ? MT : 1], x1l[HL ? MT : 1];                                                    \
    bf16x8 wa[WLDS ? 3 : 1][WLDS ? NT : 1], wb[WLDS ? 3 : 1][WLDS ? NT : 1];                                      \
    bf16x8 wal[HL ? 3 : 1][HL ? NT : 1], wbl[HL ? 3 : 1][HL ? NT : 1];                                            \
    if (v0 && v1 && v2) {                                                                                         \
      ZM_EPILOGUE((PH + 1) % 4, fz, (PH) & 1, ((PH) & 1) == 0)                                                    \
      ZM_LDX(x0, x0l, 0)                                                                                          \
      ZM_LDW(wa, wal, 0)                                                                                          \
      _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                            \
        if (s + 1 < KS) { if ((s & 1) == 0) { ZM_LDX(x1, x1l, s + 1) ZM_LDW(wb, wbl, s + 1) } else { ZM_LDX(x0, x0l, s + 1) ZM_LDW(wa, wal, s + 1) } } \
        ZM_FENCE                                                                                                  \
        if ((s & 1) == 0) { ZM_MMA((PH + 2) % 4, 2, s, x0, x0l, wa, wal) ZM_DMA(s) ZM_MMA((PH + 3) % 4, 1, s, x0, x0l, wa, wal) ZM_MMA_D0(PH, s, x0, x0l, wa, wal) }  \
        else { ZM_MMA((PH + 2) % 4, 2, s, x1, x1l, wb, wbl) ZM_DMA(s) ZM_MMA((PH + 3) % 4, 1, s, x1, x1l, wb, wbl) ZM_MMA_D0(PH, s, x1, x1l, wb, wbl) } \
      }                                                                                                           \
    } else {   /* first / last planes of a piece: guarded groups; the epilogue AFTER the loop on purpose -- placed first in */ \
               /* both branches the compiler hoists it out of them, away from the MFMAs it should hide behind */    \
      _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                            \
        ZM_LDX(x0, x0l, s)                                                                                        \
        ZM_LDW(wa, wal, s)                                                                                        \
        ZM_DMA(s)                                                                                                 \
        if (v2) { ZM_MMA((PH + 2) % 4, 2, s, x0, x0l, wa, wal) }                                                  \
        if (v1) { ZM_MMA((PH + 3) % 4, 1, s, x0, x0l, wa, wal) }                                                  \
        if (v0) { ZM_MMA_D0(PH, s, x0, x0l, wa, wal) }                                                            \
      }                                                                                                           \
      ZM_EPILOGUE((PH + 1) % 4, fz, (PH) & 1, ((PH) & 1) == 0)                                                    \
    }                                                                                                             \
    ZM_T(ts3);                                                                                                    \
    ZM_ACC(0, ts1, ts0); ZM_ACC(1, ts2, ts1); ZM_ACC(2, ts3, ts2); ZM_ACC(3, 1, 0);                               \
    if (v0 && v1 && v2) { ZM_ACC(5, ts3, ts2); ZM_ACC(6, 1, 0); }                                                 \
    ++i;                                                                                                          \
    islot = islot + 1 == NSLOT ? 0 : islot + 1;                                                                   \
  }

    // PS: one step = PT sub-steps (channel planes) of input plane i; the accumulator roles are those of ZM_STEP.  Sub-step t:
    // its plane (issued D sub-steps ago) and its weights (issued at the top of sub-step t - 1, BEFORE that sub-step's plane
    // DMAs and stores) have landed once all but the youngest NJ + NS operations are done; every sub-step issues exactly NJW + NJ
    // + NS operations (fillers / dropped stores where there is nothing to do), so the count holds from sub-step D on.
#define ZM_DMA_PS(s_) if constexpr (D == 1 && PSFRONT) { if constexpr (PSHALF) { ZM_DMA_H(s_) } } else { ZM_DMA_(s_) }
#define ZM_STEP_PS(PH)                                                                                            \
  {                                                                                                               \
    const bool v0 = i < nz, v1 = i >= 1 && i - 1 < nz, v2 = i >= 2 && i - 2 < nz;                                 \
    const int fz = (i >= 3 && i - 3 < nz) ? z0 + i - 3 : -1;                                                      \
    const int npl = i < nin ? PT : 1;      /* (the step behind the last plane carries an epilogue only) */        \
    for (int pc = 0; pc < npl; ++pc) {                                                                            \
      if (D >= 2 && tsub >= D) ZM_SYNC(NJ + NS);                                                                  \
      else ZM_SYNC(0);                                                                                            \
      {      /* the next sub-step's weights first ... */                                                         \
        int pn = pc + 1, in_ = i;                                                                                 \
        if (pn >= PT) { pn = 0; ++in_; }                                                                          \
        w_dma(in_ < nin ? pn : -1, wbuf ^ 1, false);      /* (with the memory clobber: the epilogue's stores -- whose data are */ \
                                                          /* ready at the top of the sub-step -- must not be hoisted above them)  */ \
      }                                                                                                           \
      {      /* ... then the plane D sub-steps ahead, piecewise between the MFMAs below */                       \
        int pa = pc + D, ia = i;                                                                                  \
        while (pa >= PT) { pa -= PT; ++ia; }                                                                      \
        pl_fill = false;                                                                                          \
        if (ia < nin) plane_begin(ia, (tslot + D) % NSLOT, pa);                                                   \
        else { pl_mask = 0; pl_fill = true; pl_dst0 = ring + NSLOT * S + wave * 1024; }                           \
        /* two slots (prefetch distance 1): the whole plane goes out NOW -- it has to land within this sub-step */  \
        if constexpr (D == 1 && PSFRONT && !PSHALF) { _Pragma("unroll") for (int j = 0; j < NJ; ++j) plane_dma(j, true); } \
      }                                                                                                           \
      const unsigned char* sb = ring + tslot * S;                                                                 \
      const unsigned char* wlb = wl + wbuf * WB;                                                                  \
      if (pc == 0) {      /* first contribution to output plane i */                                             \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                            \
          _Pragma("unroll") for (int m = 0; m < MT; ++m) acc[PH][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};               \
      }                                                                                                           \
      bf16x8 x0[MT], x1[MT], x0l[HL ? MT : 1], x1l[HL ? MT : 1];                                                  \
      bf16x8 wa[3][NT], wb[3][NT], wal[HL ? 3 : 1][HL ? NT : 1], wbl[HL ? 3 : 1][HL ? NT : 1];                    \
      /* the epilogue of plane i - 3 rides in the first sub-step; the others issue as many dropped stores (counted waits) */ \
      const zm_u32x2 zz_ = {0u, 0u};                                                                              \
      if (PSPIPE && v0 && v1 && v2) {      /* fragment reads of K step s + 1 in flight under the MFMAs of step s, as ZM_STEP's fast path */ \
        if (pc == 0) { ZM_EPILOGUE((PH + 1) % 4, fz, 0, false) }                                                  \
        else { _Pragma("unroll") for (int k = 0; k < NS; ++k) __builtin_amdgcn_raw_buffer_store_b64(zz_, yrs, 0x80000000u, 0, 0); } \
        ZM_LDX(x0, x0l, 0)                                                                                        \
        ZM_LDW_B(wa, wal, 0, wlb)                                                                                 \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                          \
          if (s + 1 < KS) { if ((s & 1) == 0) { ZM_LDX(x1, x1l, s + 1) ZM_LDW_B(wb, wbl, s + 1, wlb) } else { ZM_LDX(x0, x0l, s + 1) ZM_LDW_B(wa, wal, s + 1, wlb) } } \
          if ((s & 1) == 0) { ZM_MMA((PH + 2) % 4, 2, s, x0, x0l, wa, wal) ZM_DMA_PS(s) ZM_MMA((PH + 3) % 4, 1, s, x0, x0l, wa, wal) ZM_MMA(PH, 0, s, x0, x0l, wa, wal) } \
          else { ZM_MMA((PH + 2) % 4, 2, s, x1, x1l, wb, wbl) ZM_DMA_PS(s) ZM_MMA((PH + 3) % 4, 1, s, x1, x1l, wb, wbl) ZM_MMA(PH, 0, s, x1, x1l, wb, wbl) } \
        }                                                                                                         \
      } else {                                                                                                    \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                          \
          ZM_LDX(x0, x0l, s)                                                                                      \
          ZM_LDW_B(wa, wal, s, wlb)                                                                               \
          ZM_DMA_PS(s)                                                                                               \
          if (v2) { ZM_MMA((PH + 2) % 4, 2, s, x0, x0l, wa, wal) }                                                \
          if (v1) { ZM_MMA((PH + 3) % 4, 1, s, x0, x0l, wa, wal) }                                                \
          if (v0) { ZM_MMA(PH, 0, s, x0, x0l, wa, wal) }                                                          \
        }                                                                                                         \
        if (pc == 0) { ZM_EPILOGUE((PH + 1) % 4, fz, 0, false) }                                                  \
        else { _Pragma("unroll") for (int k = 0; k < NS; ++k) __builtin_amdgcn_raw_buffer_store_b64(zz_, yrs, 0x80000000u, 0, 0); } \
      }                                                                                                           \
      ++tsub;                                                                                                     \
      tslot = tslot + 1 == NSLOT ? 0 : tslot + 1;                                                                 \
      wbuf ^= 1;                                                                                                  \
    }                                                                                                             \
    ++i;                                                                                                          \
  }

    // steps 0 .. nin: the last one (no input plane left: all groups off) only carries the epilogue of the last output plane
    int i = 0, islot = 0;
    int tsub = 0, tslot = 0, wbuf = 0;      // PS: sub-step counter, its ring slot and weight buffer
    if constexpr (PS) {
      while (true) {
        ZM_STEP_PS(0)
        if (i > nin) break;
        ZM_STEP_PS(1)
        if (i > nin) break;
        ZM_STEP_PS(2)
        if (i > nin) break;
        ZM_STEP_PS(3)
        if (i > nin) break;
      }
    } else
    while (true) {
      ZM_STEP(0)
      if (i > nin) break;
      ZM_STEP(1)
      if (i > nin) break;
      ZM_STEP(2)
      if (i > nin) break;
      ZM_STEP(3)
      if (i > nin) break;
    }
#undef ZM_STEP_PS
#undef ZM_DMA_PS
#undef ZM_LDW_B
#undef ZM_STEP
#undef ZM_MMA_D0
#undef ZM_MMA0
#undef ZM_MMA
#undef ZM_MMA_X
#undef ZM_W
#undef ZM_DMA
#undef ZM_DMA_
#undef ZM_DMA_H
#undef ZM_LDW
#undef ZM_LDX
#undef ZM_EPILOGUE
    // (accumulator sets of planes beyond nz hold partial sums that belong to the next piece: never stored)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no DMA may still be landing when the statistics reuse LDS
#ifdef SP_ZM_STAMPS
  {
    ZM_T(t_end);
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    zm_sum[4] = t_end - t_begin;
    zm_sum[7] = rt_end - rt_begin;       // 100 MHz ticks
    if (lane == 0 && blockIdx.x < 1024)
      for (int k = 0; k < 8; ++k) sp_zm_stamp_buf[blockIdx.x][wave][k] = zm_sum[k];
  }
#endif
  if (STATS && stats_sl != nullptr && cur_g >= 0) {
    __syncthreads();
    flush_stats(cur_g);
  }
}

template <int P, int NT, int MT, int NSLOT, bool WLDS, int NW, int STATS, int ACT, bool HL = false, bool POOL = false, bool SPLIT = false, bool PS = false>
static int launch_zm2(const sp_conv_args* a, const void* zeros, hipStream_t st) {
  constexpr int KS = (18 * P + 3) / 4;
  constexpr int NCH = (HL ? 2 : 1) * P * (NW * MT + 2) * 18 * 2;
  constexpr int NJ = (NCH + 64 * NW - 1) / (64 * NW);
  constexpr int S = NJ * NW * 1024;
  // ring (+ 1 KiB per wave where the counted-wait filler DMAs land) (+ the weight fragments; pairs: hi and lo)
  // (+ the bias table of the ELU instances: at most 75 border classes)
  constexpr int lds_bytes = NSLOT * S + NW * 1024 + (WLDS ? (PS ? 2 : 1) * (HL ? 2 : 1) * 3 * KS * NT * 1024 : 0) + (ACT == 2 ? 75 * NT * 64 : 0);
  static_assert(lds_bytes <= 160 * 1024, "ring + weights do not fit LDS");
  ConvZmDev Q;
  Q.a = *a;
  Q.zeros = zeros;
  // tile of the flattened M index (a->ITW = tw + 2, a->TH = th); 0 / unset: NW MT rows of 16
  Q.tw = a->ITW > 2 ? a->ITW - 2 : 16;
  Q.th = a->ITW > 2 ? a->TH : NW * MT;
  Q.itw = Q.tw + 2;
  SP_CHECK_ARG(Q.tw >= 1 && Q.th >= 1 && Q.tw * Q.th <= 16 * NW * MT && Q.itw * (Q.th + 2) <= (NW * MT + 2) * 18 && Q.th + 2 <= 255 && Q.itw <= 255,
               "sp_conv3d_zm: tile %d x %d does not fit the instance (%d voxels, halo tile %d)", Q.th, Q.tw, 16 * NW * MT, (NW * MT + 2) * 18);
  Q.d_tw = make_fastdiv(Q.tw);
  Q.d_itw = make_fastdiv(Q.itw);
  Q.ntx = (a->Wo + Q.tw - 1) / Q.tw;
  Q.nty = (a->Ho + Q.th - 1) / Q.th;
  Q.ncols = (uint32_t)(a->B * Q.nty * Q.ntx);
  Q.d_tx = make_fastdiv(Q.ntx);
  Q.d_ty = make_fastdiv(Q.nty);
  const uint64_t planes = (uint64_t)Q.ncols * a->Do;
  static const int slots_env_ = getenv("SP_ZM_SLOTS") ? atoi(getenv("SP_ZM_SLOTS")) : 0;
  const int slots = slots_env_ > 0 ? slots_env_ : 256;               // resident workgroups: one per CU
  unsigned grid = planes / 4 < (uint64_t)slots ? (unsigned)(planes / 4 > 0 ? planes / 4 : 1) : (unsigned)slots;
  if (a->nslices > 1) {      // teams of nslices workgroups per XCD (see the kernel)
    grid = 8u * (unsigned)a->nslices * (32u / (unsigned)a->nslices);
    SP_CHECK_ARG(planes >= (uint64_t)grid / a->nslices, "sp_conv3d_zm: too few (column, plane) pairs for %d slices in one launch", a->nslices);
  }
  if constexpr (PS) {
    auto kern = conv_zm3_kernel<P, NT, MT, NSLOT, WLDS, NW, STATS, ACT, bf16_t, false, HL, false, false, true>;
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
  } else if constexpr (SPLIT) {
    auto kern = conv_zm3_kernel<P, NT, MT, NSLOT, WLDS, NW, STATS, ACT, bf16_t, false, false, false, true>;
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
  } else if constexpr (POOL) {
    SP_CHECK_ARG(Q.tw == 16 && Q.th == NW * MT && a->dtype_out != SP_F32 && !a->y8, "sp_conv3d_zm: the pooling epilogue needs the classic %d x 16 tile and a 16-bit output", NW * MT);
    auto kern = conv_zm3_kernel<P, NT, MT, NSLOT, WLDS, NW, STATS, ACT, bf16_t, false, HL, true>;
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
  } else if constexpr (HL) {
    auto kern = conv_zm3_kernel<P, NT, MT, NSLOT, WLDS, NW, STATS, ACT, bf16_t, false, true>;
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
  } else if (a->dtype_out == SP_F32) {
    if constexpr (ACT == 2 || STATS >= 2) {
      sp_set_error("sp_conv3d_zm: the ELU epilogue and the BatchNorm-backward epilogues are built for bf16 outputs");
      return SP_EINVAL;
    } else {
      auto kern = conv_zm3_kernel<P, NT, MT, NSLOT, WLDS, NW, STATS, ACT, float>;
      SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm");
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
    }
  } else if (a->y8) {
    if constexpr (ACT == 1 && P == 1 && NT == 2) {      // the first layer of an fp8-mode network (2 -> 32 channels)
      auto kern = conv_zm3_kernel<P, NT, MT, NSLOT, WLDS, NW, STATS, ACT, bf16_t, true>;
      SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm");
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
    } else {
      sp_set_error("sp_conv3d_zm: the e4m3 copy (y8) is built for the P=1 NT=2 forward instance (bias + LeakyReLU / identity epilogue)");
      return SP_EINVAL;
    }
  } else {
    auto kern = conv_zm3_kernel<P, NT, MT, NSLOT, WLDS, NW, STATS, ACT, bf16_t>;
    SP_ENSURE_LDS(kern, lds_bytes, "sp_conv3d_zm");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, st, Q);
  }
  SP_CHECK_LAUNCH("sp_conv3d_zm");
  return SP_OK;
}

#ifdef SP_ZM_PROBE      // tools/kres.py -DSP_ZM_PROBE: resource usage of a few instances without compiling all of them
extern "C" int sp_zm_probe(const sp_conv_args* a, const void* z, hipStream_t st) {
  return launch_zm2<2, 2, 2, 3, true, 8, 1, 1, false, true>(a, z, st) + launch_zm2<1, 1, 4, 3, true, 8, 1, 1, true, true>(a, z, st) + launch_zm2<2, 2, 2, 3, true, 8, 1, 1>(a, z, st);
}
#else
template <int P, int NT, int MT, int NSLOT, bool WLDS, int NW>
static int launch_zm(const sp_conv_args* a, const void* zeros, hipStream_t st) {
  const bool plain = a->act == SP_ACT_NONE && a->bias == nullptr;      // data gradients: nothing to do but round and store
  if (a->stats_mode == 1 || a->stats_mode == 2) {      // data gradient + (sum g, sum g x) for the BatchNorm backward (checked by the caller: plain, stats, aux), or
    if constexpr (NW == 8 && P <= 2 && MT * NT == 4 && NSLOT == 3) {      // with the whole BatchNorm / activation backward in its epilogue (dz out)
      if (a->stats_mode == 2) return launch_zm2<P, NT, MT, NSLOT, true, NW, 3, 0>(a, zeros, st);
      return launch_zm2<P, NT, MT, NSLOT, true, NW, 2, 0>(a, zeros, st);
    } else {
      sp_set_error("sp_conv3d_zm: no BatchNorm-backward instance for P=%d NT=%d NW=%d", P, NT, NW);
      return SP_EINVAL;
    }
  }
  if (a->act == SP_ACT_ELU) {
    if constexpr (NW == 8 && P <= 2) {      // the CAE's 16 / 24 / 32-channel layers
      // (with statistics the register-weight form of (1, 1) is past 256 registers: its weights in LDS)
      if (a->stats) return launch_zm2<P, NT, MT, NSLOT, true, NW, 1, 2>(a, zeros, st);
      return launch_zm2<P, NT, MT, NSLOT, WLDS, NW, 0, 2>(a, zeros, st);
    } else {
      sp_set_error("sp_conv3d_zm: no ELU instance for P=%d NW=%d", P, NW);
      return SP_EINVAL;
    }
  }
  if (a->pool_y) {      // MaxPool3d(2) of the output in the epilogue, statistics of the pooled tensor
    if constexpr (NW == 8 && NSLOT == 3 && ((P == 1 && NT == 1 && MT == 4) || (P == 2 && NT == 2 && MT == 2))) {
      return launch_zm2<P, NT, MT, NSLOT, WLDS, NW, 1, 1, false, true>(a, zeros, st);
    } else {
      sp_set_error("sp_conv3d_zm: no pooling-epilogue instance for P=%d NT=%d NW=%d", P, NT, NW);
      return SP_EINVAL;
    }
  }
  if (a->y2) {      // the data gradient of a concatenating layer: two dense output tensors from one launch
    if constexpr (NW == 8 && P == 1 && NT == 3) {
      return launch_zm2<P, NT, MT, NSLOT, WLDS, NW, 0, 0, false, false, true>(a, zeros, st);
    } else {
      sp_set_error("sp_conv3d_zm: no split-output instance for P=%d NT=%d NW=%d", P, NT, NW);
      return SP_EINVAL;
    }
  }
  if (a->stats) return launch_zm2<P, NT, MT, NSLOT, WLDS, NW, 1, 1>(a, zeros, st);
  if (plain) return launch_zm2<P, NT, MT, NSLOT, WLDS, NW, 0, 0>(a, zeros, st);
  return launch_zm2<P, NT, MT, NSLOT, WLDS, NW, 0, 1>(a, zeros, st);
}

// (P, NT) -> rows per wave, ring slots, waves per workgroup; SP_EINVAL = no kernel.  runtime/plan.py (ZM_CONFIGS) must agree:
// tests/test_cabi.py checks it.  SP_ZM_NW=4 forces the one-wave-per-SIMD set everywhere (A/B runs; read on both sides).
// three input planes (48 -> 16): eight waves on 24 x 16 tiles and TWO ring slots (three would not fit beside the 41 KB of weight
// fragments), the prefetched plane's DMAs at the top of the step.  The four-wave form (16 x 16 tiles, three slots) spent 1.8 us of
// every 2.9 us step outside the MFMAs -- DMA issue, epilogue and barrier of a wave are not covered by a partner on its SIMD --:
// 161 -> 143 us at 4 x 92^3.  SP_ZM_31=w4 brings it back (A/B runs; read on both sides).
static bool zm_31w4() { static const bool v = getenv("SP_ZM_31") && !strcmp(getenv("SP_ZM_31"), "w4"); return v; }
static bool zm_nw4() { static const bool v = getenv("SP_ZM_NW") && atoi(getenv("SP_ZM_NW")) == 4; return v; }
extern "C" int sp_conv3d_zm_config(int32_t P, int32_t NT, int32_t* MT, int32_t* NSLOT, int32_t* NW) {
  int mt = 0, ns = 3, nw = 8;
  // eight waves (two per SIMD): one wave's epilogue / DMA / LDS instructions issue under its partner's MFMAs -- measured 12-25 %
  // faster than four waves with twice the rows each, except for three input planes, where the smaller per-wave tile makes the
  // LDS weight reads (3 per 6 MFMAs) the limit and the register budget of two waves per SIMD spills
  if (P == 1 && NT == 1) mt = 4;
  else if (P == 1 && NT == 2) mt = 2;
  else if (P == 1 && NT == 3) mt = 2;
  else if (P == 2 && NT == 1) mt = 4;
  else if (P == 2 && NT == 2) mt = 2;
  else if (P == 3 && NT == 1) { mt = 3; ns = 2; if (zm_31w4() || zm_nw4()) { mt = 4; ns = 3; nw = 4; } }
  if (zm_nw4() && mt && nw == 8) { mt *= 2; nw = 4; }
  if (MT) *MT = mt;
  if (NSLOT) *NSLOT = ns;
  if (NW) *NW = nw;
  return mt ? SP_OK : SP_EINVAL;
}

// bf16 PAIR instances (dtype_in = dtype_out = SP_HL: the forward convolutions of the "bf16x3" precision mode): twice the planes in
// the ring and hi + lo weight fragments in LDS, so the tiles are smaller where Cin * Cout grows -- (P, NT) -> (MT, NSLOT, NW);
// runtime/plan.py (ZM_CONFIGS_HL) must agree (tests/test_cabi.py)
extern "C" int sp_conv3d_zm_config_hl(int32_t P, int32_t NT, int32_t* MT, int32_t* NSLOT, int32_t* NW) {
  int mt = 0, ns = 3, nw = 8;
  if (P == 1 && NT == 1) mt = 4;                               // 158 KiB: ring 3 x 40, weights 30
  else if (P == 1 && NT == 2) mt = 2;                          // 140 KiB: ring 3 x 24, weights 60
  else if (P == 2 && NT == 1) { mt = 4; ns = 2; nw = 4; }      // 146 KiB: ring 2 x 44, weights 54
  else if (P == 2 && NT == 2) { mt = 2; ns = 2; nw = 4; }      // 160 KiB: ring 2 x 24, weights 108
  else if (P == 3 && NT == 1) { mt = 2; ns = 2; nw = 4; }      // 160 KiB: ring 2 x 36, weights 84
  if (MT) *MT = mt;
  if (NSLOT) *NSLOT = ns;
  if (NW) *NW = nw;
  return mt ? SP_OK : SP_EINVAL;
}

// plane-serial instances (sp_conv_args.pser_planes > 0): output tiles NT, bf16 or bf16 pairs -> (MT, NSLOT, NW); runtime/plan.py
// (ZM_CONFIGS_PS) must agree (tests/test_cabi.py)
extern "C" int sp_conv3d_zm_config_ps(int32_t NT, int32_t hl, int32_t* MT, int32_t* NSLOT, int32_t* NW) {
  int mt = 0, ns = 3, nw = 8;
  if (!hl && NT == 1) mt = 4;                      // 110 KiB: ring 3 x 24, weight buffers 2 x 15
  else if (!hl && NT == 2) mt = 2;                 // 116 KiB: ring 3 x 16, weight buffers 2 x 30
  else if (hl && NT == 1) { mt = 4; ns = 2; }      // 148 KiB: ring 2 x 40, weight buffers 2 x 30
  if (MT) *MT = mt;
  if (NSLOT) *NSLOT = ns;
  if (NW) *NW = nw;
  return mt ? SP_OK : SP_EINVAL;
}
template <int NT, int MT, int NSLOT, bool HL>
static int launch_zm_ps(const sp_conv_args* a, const void* zeros, hipStream_t st) {
  if (a->stats) return launch_zm2<1, NT, MT, NSLOT, true, 8, 1, 1, HL, false, false, true>(a, zeros, st);
  return launch_zm2<1, NT, MT, NSLOT, true, 8, 0, 1, HL, false, false, true>(a, zeros, st);
}

template <int P, int NT, int MT, int NSLOT, int NW>
static int launch_zm_hl(const sp_conv_args* a, const void* zeros, hipStream_t st) {
  if (a->pool_y) {      // MaxPool3d(2) of the pair values in the epilogue
    if constexpr ((P == 1 && NT == 1 && MT == 4) || (P == 2 && NT == 2 && MT == 2)) {
      return launch_zm2<P, NT, MT, NSLOT, true, NW, 1, 1, true, true>(a, zeros, st);
    } else {
      sp_set_error("sp_conv3d_zm: no pooling-epilogue pair instance for P=%d NT=%d", P, NT);
      return SP_EINVAL;
    }
  }
  if (a->stats) return launch_zm2<P, NT, MT, NSLOT, true, NW, 1, 1, true>(a, zeros, st);
  return launch_zm2<P, NT, MT, NSLOT, true, NW, 0, 1, true>(a, zeros, st);
}

extern "C" int sp_conv3d_zm(const sp_conv_args* a, const void* zeros, sp_stream_t stream) {
  SP_CHECK_ARG(a && a->x && a->y && a->wfrag_hi && a->ktab && zeros, "sp_conv3d_zm: null pointer");
  const bool hl = a->dtype_in == SP_HL;
  SP_CHECK_ARG((a->dtype_in == SP_BF16 || hl) && a->in_scale == nullptr && (a->stats_mode >= 0 && a->stats_mode <= 2), "sp_conv3d_zm: bf16 (or bf16 pair) input, no affine on load");
  SP_CHECK_ARG(a->stats_mode != 1 || (a->stats && a->aux && a->act == SP_ACT_NONE && a->bias == nullptr && a->dtype_out == SP_BF16 && !hl && !a->y8 && a->nslices <= 1),
               "sp_conv3d_zm: stats_mode 1 (sum g, sum g x) is for plain bf16 data gradients with statistics rows and the layer input (aux)");
  // stats_mode 2: act / act_param describe the FIRST convolution's activation (its derivative from x = aux), not an epilogue of this one
  SP_CHECK_ARG(a->stats_mode != 2 || (a->aux && a->dz_sums && a->bnb.sums && a->bnb.gamma && a->bnb.mean && a->bnb.invstd && a->bnb.count > 0 && a->bnb.nrep >= 1 &&
                                      a->bnb.C <= a->Cout && a->bnb.CP >= a->bnb.C && (a->act == SP_ACT_LEAKY || a->act == SP_ACT_NONE) && a->bias == nullptr &&
                                      a->dtype_out == SP_BF16 && !hl && !a->y8 && a->nslices <= 1 && a->group_batch == 0 && a->stats == nullptr),
               "sp_conv3d_zm: stats_mode 2 (dz = BatchNorm / LeakyReLU backward of the data gradient) needs aux, dz_sums, bnb and a plain bf16 data gradient");
  SP_CHECK_ARG(!hl || (a->dtype_out == SP_HL && a->wfrag_lo && a->x_lo_delta != 0 && a->y_lo_delta != 0 && a->x_lo_delta % 16 == 0 && a->y_lo_delta % 8 == 0 &&
                       (a->act == SP_ACT_LEAKY || a->act == SP_ACT_NONE) && !a->y8 && a->nslices <= 1),
               "sp_conv3d_zm: bf16 pairs in -> bf16 pairs out with hi and lo weight fragments, bias + LeakyReLU / identity epilogue");
  SP_CHECK_ARG(a->sD == 1 && a->sH == 1 && a->sW == 1, "sp_conv3d_zm: stride 1 only");
  SP_CHECK_ARG(!a->y2 || (a->split_nt >= 1 && a->split_nt < a->NT && a->CPo2 >= (a->NT - a->split_nt) * 16 && a->CPo >= a->split_nt * 16 && a->act == SP_ACT_NONE && a->bias == nullptr &&
                         !a->stats && a->stats_mode == 0 && a->dtype_out == SP_BF16 && !hl && !a->y8 && !a->pool_y && a->nslices <= 1 && a->group_batch == 0 &&
                         a->osD == 1 && a->osH == 1 && a->osW == 1 && a->ooD == 0 && a->ooH == 0 && a->ooW == 0 &&
                         (uint64_t)a->YD * a->YH * a->YW * a->CPo2 * 2 < (1ull << 31)),
               "sp_conv3d_zm: y2 (split output) is for plain dense bf16 data gradients: tiles [0, split_nt) to y, the rest to y2 (pitch CPo2)");
  SP_CHECK_ARG(!a->pool_y || (a->stats && a->stats_mode == 0 && a->act != SP_ACT_ELU && a->group_batch == 0 && a->nslices <= 1 && !a->y8 && !a->bias_tab &&
                              a->osD == 1 && a->osH == 1 && a->osW == 1 && a->ooD == 0 && a->ooH == 0 && a->ooW == 0 && a->YD == a->Do && a->YH == a->Ho && a->YW == a->Wo &&
                              a->YD >= 2 && a->YH >= 2 && a->YW >= 2 && (!hl || (a->pool_lo_delta != 0 && a->pool_lo_delta % 8 == 0)) &&
                              (uint64_t)(a->YD / 2) * (a->YH / 2) * (a->YW / 2) * a->CPo * 2 < (1ull << 31)),
               "sp_conv3d_zm: pool_y (MaxPool3d(2) in the epilogue) needs a dense forward layer with statistics (of the pooled tensor), LeakyReLU / identity");
  if (a->bias_tab || a->wfrag_gstride) {
    const int pz = -a->o0D, py = -a->o0H, px = -a->o0W;
    SP_CHECK_ARG(a->act == SP_ACT_ELU && !hl && a->nslices <= 1 && a->dtype_out == SP_BF16, "sp_conv3d_zm: bias table / per-group fragments are for the bf16 ELU instances");
    SP_CHECK_ARG(!a->bias_tab || (pz >= 0 && pz <= 2 && py >= 0 && py <= 2 && px >= 0 && px <= 2 && (2 * pz + 1) * (2 * py + 1) * (2 * px + 1) <= 75 &&
                                  a->bias_tab_gstride >= (2 * pz + 1) * (2 * py + 1) * (2 * px + 1) * a->CPo),
                 "sp_conv3d_zm: bias table: padding (%d, %d, %d) gives more than 75 classes or bias_tab_gstride %d is too small", pz, py, px, a->bias_tab_gstride);
    SP_CHECK_ARG(a->wfrag_gstride == 0 || (a->wfrag_gstride > 0 && a->wfrag_gstride % 16 == 0 && a->group_batch > 0), "sp_conv3d_zm: wfrag_gstride needs BatchNorm groups (group_batch)");
  }
  SP_CHECK_ARG(a->group_batch >= 0 && (a->group_batch == 0 || (a->B % a->group_batch == 0 && a->nslices <= 1)), "sp_conv3d_zm: group_batch %d must divide the batch %d (no slices)", a->group_batch, a->B);
  SP_CHECK_ARG(a->act == SP_ACT_LEAKY || a->act == SP_ACT_NONE || a->act == SP_ACT_ELU, "sp_conv3d_zm: LeakyReLU, ELU or identity epilogue");
  SP_CHECK_ARG(a->CPi % 16 == 0 && a->NT == a->NTtot && a->Cout == 16 * a->NT && a->CPo >= (a->y2 ? 16 * a->split_nt : a->Cout), "sp_conv3d_zm: whole 16-channel tiles (CPi %d, Cout %d, NT %d)", a->CPi, a->Cout, a->NT);
  SP_CHECK_ARG(a->nslices >= 0 && a->nslices <= 16 && (a->nslices <= 1 || (a->CPo >= a->nslices * a->Cout && a->slice_wfrag_stride > 0 && a->slice_wfrag_stride % 16 == 0 && !a->y8)),
               "sp_conv3d_zm: nslices %d (CPo %d, Cout %d per slice, slice_wfrag_stride %lld; no e4m3 copy)", a->nslices, a->CPo, a->Cout, (long long)a->slice_wfrag_stride);
  SP_CHECK_ARG(!a->stats || (a->stats_nrep >= 1 && (a->stats_nrep & (a->stats_nrep - 1)) == 0), "sp_conv3d_zm: stats_nrep must be a power of two");
  SP_CHECK_ARG(a->Do > 0 && a->Ho > 0 && a->Wo > 0 && a->B > 0, "sp_conv3d_zm: empty output");
  SP_CHECK_ARG(!a->y8 || (a->dtype_out == SP_BF16 && a->osD == 1 && a->osH == 1 && a->osW == 1 && a->ooD == 0 && a->ooH == 0 && a->ooW == 0 &&
                          a->YD == a->Do && a->YH == a->Ho && a->YW == a->Wo && a->y8_scale > 0.f &&
                          (uint64_t)a->y8_plane >= (uint64_t)a->B * a->YD * a->YH * a->YW * 16),
               "sp_conv3d_zm: y8 needs a dense bf16 output, a positive y8_scale and y8_plane >= one plane of the output");
  const int P = a->CPi / 16;
  SP_CHECK_ARG(hl || a->dtype_out != SP_HL, "sp_conv3d_zm: a bf16 pair output needs a bf16 pair input");
  SP_CHECK_ARG(a->pser_planes == 0 || (a->pser_planes == P && P >= 2 && (a->act == SP_ACT_LEAKY || a->act == SP_ACT_NONE) && !a->y8 && !a->y2 && !a->pool_y && a->stats_mode == 0 &&
                                       a->nslices <= 1 && a->group_batch == 0 && !a->bias_tab && a->dtype_out != SP_F32),
               "sp_conv3d_zm: pser_planes (plane-serial march) = CPi / 16 >= 2, bias / LeakyReLU / identity epilogue");
  // every byte offset the kernel forms must fit 32 bits (per-sample base is 64-bit)
  const uint64_t span = a->x_plane ? (uint64_t)P * (uint64_t)a->x_plane * 2 : (uint64_t)a->Di * a->Hi * a->Wi * a->CPi * 2;
  SP_CHECK_ARG(span < (1ull << 31), "sp_conv3d_zm: input too large for 32-bit offsets");
  SP_CHECK_ARG((uint64_t)a->YD * a->YH * a->YW * a->CPo * 4 < (1ull << 31), "sp_conv3d_zm: output sample too large for a buffer descriptor");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int32_t mt = 0, ns = 0, nw = 0;
  if (a->pser_planes > 0) {
    SP_CHECK_ARG(sp_conv3d_zm_config_ps(a->NT, hl ? 1 : 0, &mt, &ns, &nw) == SP_OK && mt == a->MT, "sp_conv3d_zm: no plane-serial kernel for NT=%d MT=%d (%s)", a->NT, a->MT, hl ? "pairs" : "bf16");
    if (hl) return launch_zm_ps<1, 4, 2, true>(a, zeros, st);
    if (a->NT == 1) return launch_zm_ps<1, 4, 3, false>(a, zeros, st);
    return launch_zm_ps<2, 2, 3, false>(a, zeros, st);
  }
  if (hl) {
    SP_CHECK_ARG(sp_conv3d_zm_config_hl(P, a->NT, &mt, &ns, &nw) == SP_OK && mt == a->MT, "sp_conv3d_zm: no bf16-pair kernel for P=%d NT=%d MT=%d", P, a->NT, a->MT);
    if (P == 1 && a->NT == 1) return launch_zm_hl<1, 1, 4, 3, 8>(a, zeros, st);
    if (P == 1 && a->NT == 2) return launch_zm_hl<1, 2, 2, 3, 8>(a, zeros, st);
    if (P == 2 && a->NT == 1) return launch_zm_hl<2, 1, 4, 2, 4>(a, zeros, st);
    if (P == 2 && a->NT == 2) return launch_zm_hl<2, 2, 2, 2, 4>(a, zeros, st);
    if (P == 3 && a->NT == 1) return launch_zm_hl<3, 1, 2, 2, 4>(a, zeros, st);
    return SP_EINVAL;
  }
  SP_CHECK_ARG(sp_conv3d_zm_config(P, a->NT, &mt, &ns, &nw) == SP_OK && mt == a->MT, "sp_conv3d_zm: no kernel for P=%d NT=%d MT=%d", P, a->NT, a->MT);
  // SP_ZM_VARIANT=<digit per (P,NT) class in the order 11 12 13 21 22 31>: tuning knob (tools/bench_conv.py)
  static const char* var_ = getenv("SP_ZM_VARIANT");
  auto v = [&](int k) { return (var_ && (int)strlen(var_) > k) ? var_[k] - '0' : 0; };
  if (nw == 4) {
    if (P == 1 && a->NT == 1) return launch_zm<1, 1, 8, 3, false, 4>(a, zeros, st);
    if (P == 1 && a->NT == 2) return launch_zm<1, 2, 4, 3, true, 4>(a, zeros, st);
    if (P == 1 && a->NT == 3) return launch_zm<1, 3, 4, 3, true, 4>(a, zeros, st);      // (register weights: 180 + 192 accumulator registers spill)
    if (P == 2 && a->NT == 1) return launch_zm<2, 1, 8, 3, true, 4>(a, zeros, st);
    if (P == 2 && a->NT == 2) return launch_zm<2, 2, 4, 3, true, 4>(a, zeros, st);
    if (P == 3 && a->NT == 1) return v(5) == 1 ? launch_zm<3, 1, 4, 3, false, 4>(a, zeros, st) : launch_zm<3, 1, 4, 3, true, 4>(a, zeros, st);
    return SP_EINVAL;
  }
  if (P == 3 && a->NT == 1) return launch_zm<3, 1, 3, 2, true, 8>(a, zeros, st);      // (the only eight-wave instance with two ring slots)
  if (P == 1 && a->NT == 1) return v(0) == 1 ? launch_zm<1, 1, 4, 3, true, 8>(a, zeros, st) : launch_zm<1, 1, 4, 3, false, 8>(a, zeros, st);
  if (P == 1 && a->NT == 2) return launch_zm<1, 2, 2, 3, true, 8>(a, zeros, st);
  if (P == 1 && a->NT == 3) return launch_zm<1, 3, 2, 3, true, 8>(a, zeros, st);
  if (P == 2 && a->NT == 1) return launch_zm<2, 1, 4, 3, true, 8>(a, zeros, st);
  if (P == 2 && a->NT == 2) return launch_zm<2, 2, 2, 3, true, 8>(a, zeros, st);
  return SP_EINVAL;
}

#endif
