// Input-pipeline kernels (SURVEY.md 8 "next" row N4): the elastic deformation of common/data.py:313-351 on the device.
//   * scipy.ndimage.gaussian_filter(noise, sigma, mode="constant", cval=0)   (data.py:332-334)  -> sp_gaussian_filter3d
//   * scipy.ndimage.map_coordinates(image, indices, order=1)                  (data.py:339)      -> sp_map_coordinates_linear
// Volumes are C-ordered (n0, n1, n2) fp32 arrays -- the (x, y, z) numpy layout the reference's transforms work on.
// Both are HBM/L2-bound gathers over a 128 x 128 x 28 volume (1.8 MB): one thread per output element, consecutive threads
// along the contiguous axis, so every tap of every pass is a coalesced row read.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sp_common.h"

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define SP_GAUSS_MAX_RADIUS 64

struct GaussW {
  float w[2 * SP_GAUSS_MAX_RADIUS + 1];
};

// one separable pass along `axis`: out[i] = sum_k w[k] * in[i + k - r], zero outside the volume (mode="constant", cval=0)
__global__ __launch_bounds__(256) void gauss1d_kernel(const float* __restrict__ src, float* __restrict__ dst, int n0, int n1,
                                                      int n2, int axis, int radius, GaussW gw) {
  const int64_t total = (int64_t)n0 * n1 * n2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % n2);
  const int64_t r = idx / n2;
  const int j = (int)(r % n1), i = (int)(r / n1);
  const int pos = axis == 0 ? i : (axis == 1 ? j : k);
  const int len = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
  const int64_t stride = axis == 0 ? (int64_t)n1 * n2 : (axis == 1 ? n2 : 1);
  const int lo = max(-radius, -pos), hi = min(radius, len - 1 - pos);
  float acc = 0.f;
  for (int t = lo; t <= hi; ++t) acc = fmaf(gw.w[t + radius], src[idx + t * stride], acc);
  dst[idx] = acc;
}

// scipy.ndimage.gaussian_filter semantics: radius = int(truncate * sigma + 0.5), weights exp(-x^2 / (2 sigma^2)) normalised
// to sum 1 (computed in double on the host), the three axes filtered one after the other (axis 0 first); tmp: scratch of
// the volume's size; src, dst and tmp are three different buffers
extern "C" int sp_gaussian_filter3d(const float* src, float* dst, float* tmp, int32_t n0, int32_t n1, int32_t n2, float sigma,
                                    float truncate, sp_stream_t stream) {
  SP_CHECK_ARG(src && dst && tmp && tmp != src && tmp != dst && src != dst && n0 >= 1 && n1 >= 1 && n2 >= 1 && sigma > 0.f && truncate > 0.f,
               "sp_gaussian_filter3d: bad arguments");
  const int radius = (int)(truncate * sigma + 0.5f);
  SP_CHECK_ARG(radius <= SP_GAUSS_MAX_RADIUS, "sp_gaussian_filter3d: radius %d above %d", radius, SP_GAUSS_MAX_RADIUS);
  GaussW gw;
  double sum = 0.0, wd[2 * SP_GAUSS_MAX_RADIUS + 1];
  for (int t = -radius; t <= radius; ++t) { wd[t + radius] = exp(-0.5 * (double)t * t / ((double)sigma * sigma)); sum += wd[t + radius]; }
  for (int t = 0; t <= 2 * radius; ++t) gw.w[t] = (float)(wd[t] / sum);
  const int64_t total = (int64_t)n0 * n1 * n2;
  SP_CHECK_ARG(total < (1ll << 31), "sp_gaussian_filter3d: 2^31 elements or more");
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(gauss1d_kernel, dim3(grid), dim3(256), 0, ST(stream), src, dst, n0, n1, n2, 0, radius, gw);
  hipLaunchKernelGGL(gauss1d_kernel, dim3(grid), dim3(256), 0, ST(stream), (const float*)dst, tmp, n0, n1, n2, 1, radius, gw);
  hipLaunchKernelGGL(gauss1d_kernel, dim3(grid), dim3(256), 0, ST(stream), (const float*)tmp, dst, n0, n1, n2, 2, radius, gw);
  SP_CHECK_LAUNCH("sp_gaussian_filter3d");
  return SP_OK;
}

// out[i,j,k] = image sampled at (i + s0*d0[i,j,k], j + s1*d1[i,j,k], k + s2*d2[i,j,k]) -- map_coordinates(order=1,
// mode="constant", cval=0): a coordinate below 0 or above n-1 on any axis gives cval (no interpolation beyond the edges),
// otherwise trilinear interpolation between floor(c) and floor(c)+1.
__global__ __launch_bounds__(256) void warp_linear_kernel(const float* __restrict__ img, const float* __restrict__ d0,
                                                          const float* __restrict__ d1, const float* __restrict__ d2, float s0,
                                                          float s1, float s2, float cval, float* __restrict__ out, int n0,
                                                          int n1, int n2) {
  const int64_t total = (int64_t)n0 * n1 * n2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % n2);
  const int64_t r = idx / n2;
  const int j = (int)(r % n1), i = (int)(r / n1);
  const float c0 = (float)i + s0 * d0[idx], c1 = (float)j + s1 * d1[idx], c2 = (float)k + s2 * d2[idx];
  if (!(c0 >= 0.f && c0 <= (float)(n0 - 1) && c1 >= 0.f && c1 <= (float)(n1 - 1) && c2 >= 0.f && c2 <= (float)(n2 - 1))) {
    out[idx] = cval;
    return;
  }
  const float f0 = floorf(c0), f1 = floorf(c1), f2 = floorf(c2);
  const float t0 = c0 - f0, t1 = c1 - f1, t2 = c2 - f2;
  const int a0 = (int)f0, a1 = (int)f1, a2 = (int)f2;
  const int b0 = min(a0 + 1, n0 - 1), b1 = min(a1 + 1, n1 - 1), b2 = min(a2 + 1, n2 - 1);     // weight 0 when clamped
  const int64_t p00 = ((int64_t)a0 * n1 + a1) * n2, p01 = ((int64_t)a0 * n1 + b1) * n2;
  const int64_t p10 = ((int64_t)b0 * n1 + a1) * n2, p11 = ((int64_t)b0 * n1 + b1) * n2;
  const float v000 = img[p00 + a2], v001 = img[p00 + b2], v010 = img[p01 + a2], v011 = img[p01 + b2];
  const float v100 = img[p10 + a2], v101 = img[p10 + b2], v110 = img[p11 + a2], v111 = img[p11 + b2];
  const float u0 = 1.f - t0, u1 = 1.f - t1, u2 = 1.f - t2;
  out[idx] = u0 * (u1 * (u2 * v000 + t2 * v001) + t1 * (u2 * v010 + t2 * v011)) +
             t0 * (u1 * (u2 * v100 + t2 * v101) + t1 * (u2 * v110 + t2 * v111));
}

extern "C" int sp_map_coordinates_linear(const float* image, const float* d0, const float* d1, const float* d2, float s0, float s1,
                                         float s2, float cval, float* out, int32_t n0, int32_t n1, int32_t n2,
                                         sp_stream_t stream) {
  SP_CHECK_ARG(image && d0 && d1 && d2 && out && out != image && n0 >= 1 && n1 >= 1 && n2 >= 1, "sp_map_coordinates_linear: bad arguments");
  const int64_t total = (int64_t)n0 * n1 * n2;
  SP_CHECK_ARG(total < (1ll << 31), "sp_map_coordinates_linear: 2^31 elements or more");
  hipLaunchKernelGGL(warp_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), image, d0, d1, d2, s0, s1,
                     s2, cval, out, n0, n1, n2);
  SP_CHECK_LAUNCH("sp_map_coordinates_linear");
  return SP_OK;
}

// ------------------------------------------------------------------------------------------------ surface distances
// Hausdorff / average symmetric surface distance of the batch metrics (metrics.py:31-44 -> medpy 0.3.0
// metric.binary.hd / assd -> __surface_distances): border = mask XOR binary_erosion(mask, cross structure of the array's
// rank, border_value 0), distances = Euclidean distance transform of the complement of the reference border, sampled at
// the result border.  The reference hands medpy the whole (B, 1, D, H, W) tensor, so the structure and the transform
// are FIVE-dimensional: an axis of extent 1 makes every voxel a border voxel, and neighbouring batch entries are one
// voxel apart.  All of that is reproduced: arrays of rank <= 5, unit spacing.
// Exact EDT, separable: g <- min_j g[.., j, ..] + (i - j)^2 along each axis in turn.  With extents <= 128 the plain
// O(n) scan per voxel is ~270 fused min-adds per voxel for 4 x 88^3 -- less than a millisecond, no lower-envelope stack.
#define SP_SD_BIG 1.0e30f
struct Dims5 { int n[5]; };

__global__ __launch_bounds__(256) void sd_border_kernel(const float* __restrict__ x, float thr, Dims5 d, int a0, int64_t total,
                                                        float* __restrict__ border) {   // a0: first axis the array really has
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int c[5];
  int64_t r = idx;
#pragma unroll
  for (int a = 4; a >= 0; --a) { c[a] = (int)(r % d.n[a]); r /= d.n[a]; }
  const bool m = x[idx] > thr;
  bool interior = m;
  int64_t stride = 1;
#pragma unroll
  for (int a = 4; a >= 0; --a) {
    // out-of-bounds neighbours count as background (scipy binary_erosion, border_value = 0)
    if (a >= a0) {
      const bool lo = c[a] > 0 && x[idx - stride] > thr, hi = c[a] < d.n[a] - 1 && x[idx + stride] > thr;
      interior = interior && lo && hi;
    }
    stride *= d.n[a];
  }
  border[idx] = (m && !interior) ? 1.f : 0.f;
}
__global__ __launch_bounds__(256) void sd_seed_kernel(const float* __restrict__ border, int64_t total, float* __restrict__ g) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) g[idx] = border[idx] != 0.f ? 0.f : SP_SD_BIG;
}
// one axis: element (o, i, k) of an (outer, n, inner) view
__global__ __launch_bounds__(256) void sd_edt_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t total, int n,
                                                          int64_t inner) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int64_t k = idx % inner, oi = idx / inner;
  const int i = (int)(oi % n);
  const float* line = src + (oi - i) * inner + k;
  float best = SP_SD_BIG;
  for (int j = 0; j < n; ++j) {
    const float dj = (float)(i - j);
    best = fminf(best, fmaf(dj, dj, line[(int64_t)j * inner]));
  }
  dst[idx] = best;
}
// out[0] = max of g (the SQUARED distance: an exact integer, the host takes the root in double), out[1] = sum of sqrt(g)
// in double, out[2] = count, over the voxels where `at` is set
__global__ __launch_bounds__(256) void sd_stats_kernel(const float* __restrict__ g, const float* __restrict__ at, int64_t total,
                                                       double* __restrict__ out) {
  float mx = 0.f;
  double sm = 0.0, cnt = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    if (at[i] != 0.f) { const float gi = g[i]; mx = fmaxf(mx, gi); sm += sqrt((double)gi); cnt += 1.0; }
  }
  __shared__ float rm[4];
  __shared__ double rs[2][4];
  float wm = mx;
  for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
  const double ws = wave_sum_d(sm), wc = wave_sum_d(cnt);
  if ((threadIdx.x & 63) == 0) { rm[threadIdx.x >> 6] = wm; rs[0][threadIdx.x >> 6] = ws; rs[1][threadIdx.x >> 6] = wc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m4 = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
    // non-negative floats order like their bit patterns: the maximum through an integer atomic on the double's slot
    atomicMax(reinterpret_cast<unsigned long long*>(&out[0]), (unsigned long long)__float_as_uint(m4));
    atomicAdd(&out[1], rs[0][0] + rs[0][1] + rs[0][2] + rs[0][3]);
    atomicAdd(&out[2], rs[1][0] + rs[1][1] + rs[1][2] + rs[1][3]);
  }
}
__global__ void sd_finish_kernel(double* out) {      // the two maxima were accumulated as float bit patterns
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = (double)__uint_as_float((unsigned)*reinterpret_cast<unsigned long long*>(&out[0]));
    out[3] = (double)__uint_as_float((unsigned)*reinterpret_cast<unsigned long long*>(&out[3]));
  }
}

// out[6] (zeroed by the caller) = { max SQUARED distance, sum of distances, count: result-border -> reference-border;  the
// same three for reference-border -> result-border }:  hd = sqrt(max(out[0], out[3])),  assd = (out[1]/out[2] + out[4]/out[5]) / 2.
// ws: 4 * prod(dims) floats.  A mask without voxels gives count 0 (the caller decides, as medpy raises).
extern "C" int sp_surface_distances(const float* result, const float* reference, float threshold, int32_t ndim, const int32_t* dims,
                                    float* ws, double* out, sp_stream_t stream) {
  SP_CHECK_ARG(result && reference && dims && ws && out && ndim >= 1 && ndim <= 5, "sp_surface_distances: bad arguments");
  Dims5 d;
  int64_t total = 1;
  for (int a = 0; a < 5; ++a) {
    const int src = a - (5 - ndim);
    d.n[a] = src >= 0 ? dims[src] : 1;
    SP_CHECK_ARG(d.n[a] >= 1, "sp_surface_distances: empty axis");
    total *= d.n[a];
  }
  SP_CHECK_ARG(total < (1ll << 31), "sp_surface_distances: 2^31 voxels or more");
  // an axis of extent 1 that the array HAS makes every voxel a border voxel (both neighbours are out of bounds); the
  // leading axes added here to reach rank 5 must not: the border kernel tests axes a0.. only
  const int a0 = 5 - ndim;
  const unsigned grid = (unsigned)((total + 255) / 256);
  float* br = ws; float* bt = ws + total; float* g0 = ws + 2 * total; float* g1 = ws + 3 * total;
  hipStream_t st = ST(stream);
  hipLaunchKernelGGL(sd_border_kernel, dim3(grid), dim3(256), 0, st, result, threshold, d, a0, total, br);
  hipLaunchKernelGGL(sd_border_kernel, dim3(grid), dim3(256), 0, st, reference, threshold, d, a0, total, bt);
  for (int dir = 0; dir < 2; ++dir) {
    const float* seed = dir == 0 ? bt : br;
    const float* at = dir == 0 ? br : bt;
    hipLaunchKernelGGL(sd_seed_kernel, dim3(grid), dim3(256), 0, st, seed, total, g0);
    float* a_ = g0; float* b_ = g1;
    int64_t inner = 1;
    for (int a = 4; a >= 0; --a) {
      if (d.n[a] > 1) {
        hipLaunchKernelGGL(sd_edt_axis_kernel, dim3(grid), dim3(256), 0, st, (const float*)a_, b_, total, d.n[a], inner);
        float* t_ = a_; a_ = b_; b_ = t_;
      }
      inner *= d.n[a];
    }
    const unsigned sg = grid < 1024 ? grid : 1024;
    hipLaunchKernelGGL(sd_stats_kernel, dim3(sg), dim3(256), 0, st, (const float*)a_, at, total, out + 3 * dir);
  }
  hipLaunchKernelGGL(sd_finish_kernel, dim3(1), dim3(64), 0, st, out);
  SP_CHECK_LAUNCH("sp_surface_distances");
  return SP_OK;
}
