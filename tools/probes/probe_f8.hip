// Hardware probe (tools/probes): operand lane maps of v_mfma_f32_16x16x128_f8f6f4 (fp8 e4m3 / bf8 e5m2), the rate of the
// scaled / non-scaled / bf16 forms, and what ds_read_b64_tr_b8 returns.  Not part of the library; run once on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- host fp8 encoders (exact for the small integers / powers of two used here)
static unsigned char enc_e4m3(float v) {   // OCP e4m3fn, bias 7
  if (v == 0) return 0;
  unsigned char s = v < 0 ? 0x80 : 0; v = fabsf(v);
  int e; float m = frexpf(v, &e);  // v = m * 2^e, m in [0.5,1)
  e -= 1; m *= 2;                  // m in [1,2)
  int E = e + 7;
  if (E <= 0) { int q = (int)lrintf(v * 512.f); return s | (unsigned char)q; }
  int M = (int)lrintf((m - 1.f) * 8.f);
  if (M == 8) { M = 0; E++; }
  return s | (unsigned char)((E << 3) | M);
}
static unsigned char enc_e5m2(float v) {   // bias 15
  if (v == 0) return 0;
  unsigned char s = v < 0 ? 0x80 : 0; v = fabsf(v);
  int e; float m = frexpf(v, &e); e -= 1; m *= 2;
  int E = e + 15;
  int M = (int)lrintf((m - 1.f) * 4.f);
  if (M == 4) { M = 0; E++; }
  return s | (unsigned char)((E << 2) | M);
}

__global__ void mfma_layout(const unsigned char* A /*[16][128]*/, const unsigned char* Bm /*[128][16]*/, float* D, int hyp, int blgp, int sa, int sb) {
  const int l = threadIdx.x;
  unsigned char a[32], b[32];
  for (int j = 0; j < 32; ++j) {
    int k = hyp == 0 ? 32 * (l >> 4) + j : 16 * (l >> 4) + (j & 15) + 64 * (j >> 4);
    a[j] = A[(l & 15) * 128 + k];
    b[j] = Bm[k * 16 + (l & 15)];
  }
  v8i av, bv;
  memcpy(&av, a, 32); memcpy(&bv, b, 32);
  v4f c = {0, 0, 0, 0};
  if (blgp == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa, 0, sb);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 1, 0, sa, 0, sb);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void mfma_layout_ns(const unsigned char* A, const unsigned char* Bm, float* D, int blgp) {
  const int l = threadIdx.x;
  unsigned char a[32], b[32];
  for (int j = 0; j < 32; ++j) { int k = 32 * (l >> 4) + j; a[j] = A[(l & 15) * 128 + k]; b[j] = Bm[k * 16 + (l & 15)]; }
  v8i av, bv; memcpy(&av, a, 32); memcpy(&bv, b, 32);
  v4f c = {0, 0, 0, 0};
  if (blgp == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 1, 0, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// ---- rate: NACC independent accumulators, ITER iterations, operands in registers
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void rate(float* out, int iters, unsigned seed) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (int)((threadIdx.x * 2654435761u + i * 40503u + seed) & 0x3f3f3f3f) | 0x20202020; b[i] = (int)((threadIdx.x * 40503u + i * 2654435761u + seed) & 0x3f3f3f3f) | 0x20202020; }
  v4f c[NACC];
  for (int n = 0; n < NACC; ++n) c[n] = v4f{0, 0, 0, 0};
  v16f c32[NACC > 4 ? 4 : NACC];
  for (int n = 0; n < (NACC > 4 ? 4 : NACC); ++n) for (int q = 0; q < 16; ++q) c32[n][q] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
      if (MODE == 0) { v8s as, bs; memcpy(&as, &a, 16); memcpy(&bs, &b, 16); c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as, bs, c[n], 0, 0, 0); }
      if (MODE == 1) c[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[n], 0, 0, 0, 0, 0, 0);
      if (MODE == 2) c[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[n], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (MODE == 3) c[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[n], 0, 1, 0, 0, 0, 0);
      if (MODE == 4) { long al, bl; memcpy(&al, &a, 8); memcpy(&bl, &b, 8); c[n] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al, bl, c[n], 0, 0, 0); }
      if (MODE == 5) { if (n < 4) c32[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c32[n], 0, 0, 0, 0, 0, 0); }
    }
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n) s += c[n][0] + c[n][1] + c[n][2] + c[n][3];
  for (int n = 0; n < (NACC > 4 ? 4 : NACC); ++n) s += c32[n][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC>
static void run_rate(const char* name, int wg_threads, double flop_per_mfma, int nacc_eff) {
  float* out; CK(hipMalloc(&out, 256 * 8 * 512 * sizeof(float)));
  const int iters = 4096, grid = 256 * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<MODE, NACC>), dim3(grid), dim3(wg_threads), 0, 0, out, iters, 12345u + rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)grid * (wg_threads / 64) * iters * nacc_eff * flop_per_mfma;
    if (rep == 2) printf("rate %-44s wg=%d threads: %.3f ms  %.1f TFLOP/s\n", name, wg_threads, ms, flops / ms / 1e9);
  }
  CK(hipFree(out));
}

__global__ void tr8(const unsigned char* src, int* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = src[i];
  __syncthreads();
  const int l = threadIdx.x;
  int addr = 0;
  if (mode == 0) addr = l * 8;                                    // lane-linear
  if (mode == 1) addr = (l & 15) * 64 + (l >> 4) * 8;             // 16 rows of 64 bytes, 8-byte column block per 16-lane group
  if (mode == 2) addr = (l & 15) * 16 + (l >> 4) * 256;           // [vox][16 ch] rows of 16 bytes, first 8 channels; 16 voxels per group
  if (mode == 3) addr = (l & 15) * 16 + 8 + (l >> 4) * 256;       // ... channels 8..15
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + addr));
  out[l * 2] = r.x; out[l * 2 + 1] = r.y;
}

int main() {
  // ---------------- MFMA layout
  unsigned char hA[16 * 128], hB[128 * 16], hB5[128 * 16];
  float fA[16 * 128], fB[128 * 16];
  srand(7);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) { float v = (float)((rand() % 9) - 4); fA[i * 128 + k] = v; hA[i * 128 + k] = enc_e4m3(v); }
  for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) { float v = (float)((rand() % 7) - 3); fB[k * 16 + j] = v; hB[k * 16 + j] = enc_e4m3(v); hB5[k * 16 + j] = enc_e5m2(v); }
  float ref[256];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 128; ++k) s += fA[i * 128 + k] * fB[k * 16 + j]; ref[i * 16 + j] = s; }
  unsigned char *dA, *dB, *dB5; float* dD;
  CK(hipMalloc(&dA, sizeof hA)); CK(hipMalloc(&dB, sizeof hB)); CK(hipMalloc(&dB5, sizeof hB5)); CK(hipMalloc(&dD, 256 * 4));
  CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice)); CK(hipMemcpy(dB5, hB5, sizeof hB5, hipMemcpyHostToDevice));
  float hD[256];
  for (int hyp = 0; hyp < 2; ++hyp) for (int blgp = 0; blgp < 2; ++blgp) {
    hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, dA, blgp ? dB5 : dB, dD, hyp, blgp, 127, 127);
    CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("mfma_scale 16x16x128 hyp %d (0: k = 32*(lane>>4)+j) B=%s scale 127/127: %d of 256 wrong (D[0]=%g ref %g)\n", hyp, blgp ? "e5m2" : "e4m3", bad, hD[0], ref[0]);
  }
  for (int blgp = 0; blgp < 2; ++blgp) {
    hipLaunchKernelGGL(mfma_layout_ns, dim3(1), dim3(64), 0, 0, dA, blgp ? dB5 : dB, dD, blgp);
    CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("mfma NON-scaled 16x16x128 hyp 0 B=%s: %d of 256 wrong\n", blgp ? "e5m2" : "e4m3", bad);
  }
  // scales: a = 128 (x2), b = 126 (x0.5), and byte select
  { hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0, 128, 127); CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != 2 * ref[i]; printf("scale_a=128 -> x2: %d wrong\n", bad);
    hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, 0, 127, 125); CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
    bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != 0.25f * ref[i]; printf("scale_b=125 -> x0.25: %d wrong\n", bad); }
  // ---------------- rates
  run_rate<0, 8>("bf16 16x16x32 (8 acc)", 256, 2.0 * 16 * 16 * 32, 8);
  run_rate<1, 8>("fp8 16x16x128 non-scaled (8 acc)", 256, 2.0 * 16 * 16 * 128, 8);
  run_rate<2, 8>("fp8 16x16x128 scaled 0x7f (8 acc)", 256, 2.0 * 16 * 16 * 128, 8);
  run_rate<3, 8>("fp8 x bf8 16x16x128 non-scaled (8 acc)", 256, 2.0 * 16 * 16 * 128, 8);
  run_rate<4, 8>("fp8 16x16x32 _fp8_fp8 (8 acc)", 256, 2.0 * 16 * 16 * 32, 8);
  run_rate<5, 4>("fp8 32x32x64 non-scaled (4 acc)", 256, 2.0 * 32 * 32 * 64, 4);
  run_rate<1, 8>("fp8 16x16x128 non-scaled (8 acc), 2 waves/SIMD", 512, 2.0 * 16 * 16 * 128, 8);
  run_rate<0, 8>("bf16 16x16x32 (8 acc), 2 waves/SIMD", 512, 2.0 * 16 * 16 * 32, 8);
  // ---------------- ds_read_b64_tr_b8
  unsigned char hs[8192]; for (int i = 0; i < 8192; ++i) hs[i] = (unsigned char)(i & 0xff);
  unsigned char* ds_; int* dout; CK(hipMalloc(&ds_, 8192)); CK(hipMalloc(&dout, 128 * 4)); CK(hipMemcpy(ds_, hs, 8192, hipMemcpyHostToDevice));
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(tr8, dim3(1), dim3(64), 0, 0, ds_, dout, mode);
    int ho[128]; CK(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b8 mode %d (lds[a] = a & 255): lane: 8 result bytes\n", mode);
    for (int l = 0; l < 64; ++l) {
      unsigned char* p = (unsigned char*)&ho[l * 2];
      printf("  lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" %3d", p[j]); printf("\n");
    }
  }
  printf("probe done\n");
  return 0;
}
