// Data side of the training loop on the device: the torchio transforms of train.py:38-62 applied to a batch of
// raw volumes already in HBM -- RandomAffine(degrees=15, p=.5) + RandomFlip(axes=(0,), p=.5) (spatial, one resampling pass) and
// RescaleIntensity(out_min_max=(0,1)) (per-volume min/max pass + one streaming pass).  torchio (0.20.16, requirements.txt:6) is
// not installed in this image: the arithmetic below follows its published algorithm (RescaleIntensity.rescale: clip to the (0,100)
// percentiles = no-op, x -= min; x /= range; x *= out_range; x += out_min, all in float32; unchanged when range == 0) and the
// oracle (oracle/data_ref.py) is a numpy restatement of the same -- parity with torchio itself is unpinned.
// All three kernels are HBM-bound streams: 12.3 MB per (120,160,160) volume and pass.
#include "common.hpp"
#include "../../include/gaviko_hip.h"

namespace gvk {

constexpr int kMmSlabs = 256;     // partial (min, max) pairs per volume (= the block size of the kernels that reduce them)

// stage 1: slab x volume -> (min, max); float4 loads, 256 threads
__global__ __launch_bounds__(256) void volume_minmax_kernel(const float* __restrict__ x, float* __restrict__ part, long long V) {
  __shared__ float smin[256], smax[256];
  const int b = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
  const long long V4 = V >> 2;
  const long long per = (V4 + kMmSlabs - 1) / kMmSlabs;
  const long long i0 = slab * per, i1 = min(V4, i0 + per);
  const f32x4* x4 = (const f32x4*)(x + (size_t)b * V);
  float lo = INFINITY, hi = -INFINITY;
  long long i = i0 + tid;
  for (; i + 768 < i1; i += 1024) {                          // four independent 16-byte loads in flight per lane
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lo = fminf(fminf(lo, fminf(v[u][0], v[u][1])), fminf(v[u][2], v[u][3]));
      hi = fmaxf(fmaxf(hi, fmaxf(v[u][0], v[u][1])), fmaxf(v[u][2], v[u][3]));
    }
  }
  for (; i < i1; i += 256) {
    const f32x4 v = x4[i];
    lo = fminf(fminf(lo, fminf(v[0], v[1])), fminf(v[2], v[3]));
    hi = fmaxf(fmaxf(hi, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  if (slab == kMmSlabs - 1)                                  // tail elements when V is not a multiple of 4
    for (long long i = (V4 << 2) + tid; i < V; i += 256) {
      const float v = x[(size_t)b * V + i];
      lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
  smin[tid] = lo; smax[tid] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { smin[tid] = fminf(smin[tid], smin[tid + s]); smax[tid] = fmaxf(smax[tid], smax[tid + s]); }
    __syncthreads();
  }
  if (tid == 0) { part[((size_t)b * kMmSlabs + slab) * 2] = smin[0]; part[((size_t)b * kMmSlabs + slab) * 2 + 1] = smax[0]; }
}

// every thread of a 256-thread block calls this (block-uniform): thread t brings partial t, the block reduces through LDS
__device__ __forceinline__ void reduce_partials(const float* part, int b, float& lo, float& hi) {
  __shared__ float rlo[256], rhi[256];
  const int tid = threadIdx.x;
  rlo[tid] = part[((size_t)b * kMmSlabs + tid) * 2];
  rhi[tid] = part[((size_t)b * kMmSlabs + tid) * 2 + 1];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { rlo[tid] = fminf(rlo[tid], rlo[tid + s]); rhi[tid] = fmaxf(rhi[tid], rhi[tid + s]); }
    __syncthreads();
  }
  lo = rlo[0]; hi = rhi[0];
}

// stage 2: y = ((x - min) / (max - min)) * (out_max - out_min) + out_min, the four float32 steps of torchio in that order
__global__ __launch_bounds__(256) void rescale_intensity_kernel(const float* __restrict__ x, const float* __restrict__ part, float* __restrict__ y,
                                                                float* __restrict__ minmax, long long V, float out_min, float out_max) {
  const int b = blockIdx.y;
  float lo, hi;
  reduce_partials(part, b, lo, hi);
  if (minmax != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { minmax[2 * b] = lo; minmax[2 * b + 1] = hi; }
  const float range = hi - lo, out_range = out_max - out_min;
  const long long V4 = V >> 2;
  const f32x4* x4 = (const f32x4*)(x + (size_t)b * V);
  f32x4* y4 = (f32x4*)(y + (size_t)b * V);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V4; i += (long long)gridDim.x * 256) {
    f32x4 v = x4[i];
    if (range != 0.f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = v[e] - lo;
        t = t / range;
        t = t * out_range;
        v[e] = t + out_min;
      }
    }
    y4[i] = v;
  }
  if (blockIdx.x == 0)
    for (long long i = (V4 << 2) + threadIdx.x; i < V; i += 256) {
      float t = x[(size_t)b * V + i];
      if (range != 0.f) { t = t - lo; t = t / range; t = t * out_range; t = t + out_min; }
      y[(size_t)b * V + i] = t;
    }
}

// Spatial transform: out[b][z][y][x] = in[b] sampled at p = A_b . mirror(z,y,x) + t_b, trilinear, neighbours outside the volume read
// the pad value (the volume's minimum: torchio's default_pad_value='minimum').  flags[b]: bits 0..2 mirror axis 0..2 of the OUTPUT
// index (RandomFlip), bit 3 = the affine map is live (otherwise an exact gather of the mirrored index).
__global__ __launch_bounds__(256) void spatial_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ mats,
                                                      const int* __restrict__ flags, const float* __restrict__ part, int D, int H, int W) {
  const int b = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int fl = flags[b];
  float pad = 0.f, hi_unused;
  if (fl & 8) reduce_partials(part, b, pad, hi_unused);        // block-uniform (b = blockIdx.z), before any thread leaves
  if (x >= W || y >= H) return;
  const float* src = in + (size_t)b * D * H * W;
  float* dst = out + (size_t)b * D * H * W;
  const int mx = (fl & 4) ? W - 1 - x : x, my = (fl & 2) ? H - 1 - y : y;
  if (!(fl & 8)) {
    for (int z = 0; z < D; ++z) {
      const int mz = (fl & 1) ? D - 1 - z : z;
      dst[((size_t)z * H + y) * W + x] = src[((size_t)mz * H + my) * W + mx];
    }
    return;
  }
  const float* m = mats + (size_t)b * 12;                  // row-major 3x4, array-axis order (axis 0 = depth)
  for (int z = 0; z < D; ++z) {
    const float qz = (float)((fl & 1) ? D - 1 - z : z), qy = (float)my, qx = (float)mx;
    const float pz = m[0] * qz + m[1] * qy + m[2] * qx + m[3];
    const float py = m[4] * qz + m[5] * qy + m[6] * qx + m[7];
    const float px = m[8] * qz + m[9] * qy + m[10] * qx + m[11];
    const float fz = floorf(pz), fy = floorf(py), fx = floorf(px);
    const int iz = (int)fz, iy = (int)fy, ix = (int)fx;
    const float wz = pz - fz, wy = py - fy, wx = px - fx;
    float acc = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int zz = iz + dz, yy = iy + dy, xx = ix + dx;
          const bool ok = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
          const float v = ok ? src[((size_t)zz * H + yy) * W + xx] : pad;
          const float w = (dz ? wz : 1.f - wz) * (dy ? wy : 1.f - wy) * (dx ? wx : 1.f - wx);
          acc += w * v;
        }
    dst[((size_t)z * H + y) * W + x] = acc;
  }
}

}  // namespace gvk

extern "C" int gvk_volume_minmax(const float* x, float* partials, int B, int64_t V, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && partials && B > 0 && V > 0, "gvk_volume_minmax: bad arguments");
  GVK_REQUIRE(((uintptr_t)x & 15) == 0 && V % 4 == 0, "gvk_volume_minmax: volumes must be 16-byte aligned with V a multiple of 4");
  GVK_LAUNCH(volume_minmax_kernel, dim3(kMmSlabs, B), dim3(256), 0, (hipStream_t)stream, x, partials, (long long)V);
  return check_launch("volume_minmax");
}

extern "C" int gvk_minmax_partials(void) { return gvk::kMmSlabs * 2; }

extern "C" int gvk_rescale_intensity(const float* x, const float* partials, float* y, float* minmax, int B, int64_t V, float out_min, float out_max,
                                     void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x && partials && y && B > 0 && V > 0, "gvk_rescale_intensity: bad arguments");
  GVK_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && V % 4 == 0, "gvk_rescale_intensity: 16-byte aligned volumes, V a multiple of 4");
  const int gx = (int)std::min<int64_t>((V / 4 + 255) / 256, 1024);
  GVK_LAUNCH(rescale_intensity_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, partials, y, minmax, (long long)V, out_min, out_max);
  return check_launch("rescale_intensity");
}

extern "C" int gvk_spatial_transform(const float* in, float* out, const float* mats, const int32_t* flags, const float* partials, int B, int D, int H,
                                     int W, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(in && out && mats && flags && partials && in != out, "gvk_spatial_transform: null pointer or in-place call");
  GVK_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535, "gvk_spatial_transform: bad shape");
  GVK_LAUNCH(spatial_kernel, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, in, out, mats, (const int*)flags, partials, D, H, W);
  return check_launch("spatial_transform");
}
