// Hardware-semantics probe for gfx950: verifies, with exact integer data, every lane map the kernels rely on.
//   1. v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 operand + accumulator lane maps
//   2. ds_read_b64_tr_b16 (4x16 transposed LDS read)
//   3. global_load_lds_dwordx4 destination = wave base + lane*16
//   4. a 32x32 accumulator tile reused as the B operand of the next 32x32x16 MFMA (permuted k order)
// Build: hipcc --offload-arch=gfx950 -O2 probe_layouts.hip -o probe_layouts ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

// ---- 1a: 16x16x32.  A[16][32], B[32][16] row-major floats in, D[16][16] out.
__global__ void k_mfma16(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + j];      // A[row l&15][k = 8(l>>4)+j]
    b[j] = (__bf16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];    // B[k][col l&15]
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];   // row=(l>>4)*4+r, col=l&15
}
// ---- 1b: 32x32x16.  A[32][16], B[16][32], D[32][32].
__global__ void k_mfma32(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = (__bf16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// ---- 2: tr16 read.  LDS tile [16 rows][64 cols] u16 = row*256+col.  Each 16-lane group g reads the 4x16 block
// rows 4g..4g+3, cols 16..31: lane 4q+p supplies &tile[4g+q][16+4p]; expect lane i elem e == tile[4g+e][16+i].
__global__ void k_tr16(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[16][64];
  const int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) tile[i / 64][i % 64] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)&tile[4 * g + q][16 + 4 * p]);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
// ---- 3: LDS-DMA.  Each lane's source = src + perm(lane)*16 B; LDS dest should be base + lane*16.
__global__ void k_glds(const unsigned int* src, unsigned int* out) {
  __shared__ __attribute__((aligned(16))) unsigned int buf[2][256];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int perm = (l * 7 + 3) & 63;
  __builtin_amdgcn_global_load_lds((const GLB_AS void*)(src + (w * 64 + perm) * 4), (LDS_AS void*)&buf[w][0], 16, 0, 0);
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[(w * 64 + l) * 4 + e] = buf[w][l * 4 + e];
}
// ---- 4: X = A1.B1 (32x32, K=16) kept in the accumulator, then Y = A2 . X  (A2 [32][32]) using X as B operand.
__global__ void k_acc_as_b(const float* A1, const float* B1, const float* A2, float* Y) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A1[r * 16 + 8 * h + j];
    b[j] = (__bf16)B1[(8 * h + j) * 32 + r];
  }
  f32x16 x = {};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 y = {};
  for (int s = 0; s < 2; ++s) {
    bf16x8 xb, a2;
    for (int j = 0; j < 8; ++j) {
      xb[j] = (__bf16)x[8 * s + j];
      const int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);   // row of X held in register 8s+j of lane half h
      a2[j] = (__bf16)A2[r * 32 + k];
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
  }
  for (int q = 0; q < 16; ++q) Y[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = y[q];
}

static int ival(int a, int b, int m, int off) { return ((a * 7 + b * 13 + off) % m) - m / 2; }

int main() {
  int fails = 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate,
         prop.sharedMemPerBlock);
  {  // 1a
    std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = ival(i, k, 9, 1);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = ival(k, 3 * j, 7, 2);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 256 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += (D[i] != R[i]);
    printf("[1a] mfma_f32_16x16x32_bf16 lane maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // 1b
    std::vector<float> A(32 * 16), B(16 * 32), D(1024), R(1024, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = ival(i, k, 9, 1);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = ival(k, 3 * j, 7, 2);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024 * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += (D[i] != R[i]);
    printf("[1b] mfma_f32_32x32x16_bf16 lane maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // 2
    unsigned short* d; std::vector<unsigned short> o(256);
    CK(hipMalloc(&d, 512));
    k_tr16<<<1, 64>>>(d);
    CK(hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
      const int g = l >> 4, i = l & 15; const unsigned short want = (unsigned short)((4 * g + e) * 256 + 16 + i);
      bad += (o[l * 4 + e] != want);
    }
    printf("[2] ds_read_b64_tr_b16 map: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    if (bad) for (int l = 0; l < 64; l += 5) printf("   lane %2d got (r,c) = (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, o[l*4]>>8, o[l*4]&255, o[l*4+1]>>8, o[l*4+1]&255, o[l*4+2]>>8, o[l*4+2]&255, o[l*4+3]>>8, o[l*4+3]&255);
  }
  {  // 3
    std::vector<unsigned int> s(512), o(512);
    for (int i = 0; i < 512; ++i) s[i] = 0xABC00000u + i;
    unsigned int *ds, *dd; CK(hipMalloc(&ds, 2048)); CK(hipMalloc(&dd, 2048));
    CK(hipMemcpy(ds, s.data(), 2048, hipMemcpyHostToDevice));
    k_glds<<<1, 128>>>(ds, dd);
    CK(hipMemcpy(o.data(), dd, 2048, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
      const int perm = (l * 7 + 3) & 63; bad += (o[(w * 64 + l) * 4 + e] != s[(w * 64 + perm) * 4 + e]);
    }
    printf("[3] global_load_lds_dwordx4 dest = base + lane*16: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // 4
    std::vector<float> A1(32 * 16), B1(16 * 32), A2(32 * 32), X(1024, 0.f), R(1024, 0.f), Y(1024);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A1[i * 16 + k] = ival(i, k, 4, 1);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B1[k * 32 + j] = ival(k, 5 * j, 4, 3);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 32; ++k) A2[i * 32 + k] = ival(3 * i, k, 5, 2);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) X[i * 32 + j] += A1[i * 16 + k] * B1[k * 32 + j];
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 32; ++k) R[i * 32 + j] += A2[i * 32 + k] * X[k * 32 + j];
    float *d1, *d2, *d3, *dy;
    CK(hipMalloc(&d1, A1.size() * 4)); CK(hipMalloc(&d2, B1.size() * 4)); CK(hipMalloc(&d3, A2.size() * 4)); CK(hipMalloc(&dy, 4096));
    CK(hipMemcpy(d1, A1.data(), A1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, B1.data(), B1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d3, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
    k_acc_as_b<<<1, 64>>>(d1, d2, d3, dy);
    CK(hipMemcpy(Y.data(), dy, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += (Y[i] != R[i]);
    printf("[4] accumulator tile as B operand (permuted k): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  printf("probe: %s\n", fails ? "SOME FAILED" : "ALL PASS");
  return fails ? 1 : 0;
}
