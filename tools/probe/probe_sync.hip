// What does one cross-stream dependency cost on the stream that waits?  A chain of tiny kernels on stream A in which every k-th kernel
// (a) has no dependency, (b) waits for an (already complete) event of stream B, (c) waits for an event stream B records right before,
// (d) same with hipStreamWriteValue32 / hipStreamWaitValue32 memory semaphores, (e) records an event nobody waits for.
// Prints microseconds per chain link.   hipcc -O2 --offload-arch=gfx950 probe_sync.hip -o probe_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__global__ void tiny(int* p) { if (threadIdx.x == 0 && p) atomicAdd(p, 1); }
__global__ void spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }

int main() {
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  int* cnt; CK(hipMalloc(&cnt, 4)); CK(hipMemset(cnt, 0, 4));
  uint32_t* sem; CK(hipMalloc(&sem, 4096)); CK(hipMemset(sem, 0, 4096));
  const int N = 400;
  const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, flags));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto run = [&](const char* name, int mode, int busy) -> int {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(sem, 0, 4096, A));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, A));
      for (int i = 0; i < N; ++i) {
        if (busy) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, 500LL);    // 5 us of work per link
        else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, A, cnt);
        switch (mode) {
          case 0: break;
          case 1: CK(hipEventRecord(ev[i], A)); break;                                           // record only
          case 2: hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, cnt); CK(hipEventRecord(ev[i], B)); CK(hipStreamWaitEvent(A, ev[i], 0)); break;
          case 3: CK(hipEventRecord(ev[i], A)); CK(hipStreamWaitEvent(B, ev[i], 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, cnt);
                  CK(hipEventRecord(ev[(i + N / 2) % N], B)); break;                               // A -> B only (A never waits)
          case 4: hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, cnt); CK(hipStreamWriteValue32(B, sem, (uint32_t)(i + 1), 0));
                  CK(hipStreamWaitValue32(A, sem, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu)); break;
          case 5: CK(hipStreamWriteValue32(A, sem + 64, (uint32_t)(i + 1), 0)); break;            // write only
        }
      }
      CK(hipEventRecord(t1, A));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 2) printf("%-58s %s links: %7.2f us per link\n", name, busy ? "5-us" : "tiny", ms * 1e3 / N);
    }
    return 0;
  };
  for (int busy = 0; busy < 2; ++busy) {
    if (run("plain chain on one stream", 0, busy)) return 1;
    if (run("+ an event record per link (nobody waits)", 1, busy)) return 1;
    if (run("+ wait for an event another stream records per link", 2, busy)) return 1;
    if (run("+ record per link that another stream waits for", 3, busy)) return 1;
    if (run("+ wait for a memory semaphore another stream writes", 4, busy)) return 1;
    if (run("+ a memory-semaphore write per link", 5, busy)) return 1;
  }
  return 0;
}
