// Scheduling-semantics probe: do cross-stream event edges cost more than the dependency they express?
// Kernels spin on the constant 100 MHz wall clock, so durations do not depend on shared resources; only ordering shows.
// Mimics the GAViKO backward layer boundary (main: three big kernels per layer; side: 12 small ones, an early event the main
// chain waits for, and a write-after-read join before the third big kernel).  Runs eagerly and as a captured HIP graph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(long long ticks, int* sink) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 9999) sink[0] = lds[0];
}

static hipStream_t mainS, sideS, side2S;
static int LAYERS = 12;

static void big(hipStream_t s) { hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 65536, s, 5000LL, nullptr); }    // 2 rounds x 50 us
static void small_(hipStream_t s) { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, 1500LL, nullptr); }        // 15 us

static void edge(hipStream_t from, hipStream_t to) {
  hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); hipEventRecord(e, from); hipStreamWaitEvent(to, e, 0);
}

static void body(int mode) {   // 0 main only, 1 side only, 2 forked, 3 forked with a second side chain
  if (mode == 0) { for (int l = 0; l < LAYERS; ++l) { big(mainS); big(mainS); big(mainS); } return; }
  if (mode == 1) { for (int l = 0; l < LAYERS; ++l) for (int k = 0; k < 12; ++k) small_(mainS); return; }
  edge(mainS, sideS);
  if (mode == 3) edge(mainS, side2S);
  for (int l = 0; l < LAYERS; ++l) {
    for (int k = 0; k < 3; ++k) small_(sideS);
    hipEvent_t early; hipEventCreateWithFlags(&early, hipEventDisableTiming); hipEventRecord(early, sideS);
    for (int k = 0; k < 9; ++k) small_(sideS);
    big(mainS);
    hipStreamWaitEvent(mainS, early, 0);
    if (mode == 3) { hipStreamWaitEvent(side2S, early, 0); for (int k = 0; k < 14; ++k) small_(side2S); }
    big(mainS);
    edge(sideS, mainS);           // WAR join before the last kernel of the layer
    big(mainS);
    edge(mainS, sideS);           // next layer's side chain needs this layer's result
  }
  edge(sideS, mainS);
  if (mode == 3) edge(side2S, mainS);
}

template <class F> static double timeit(F f, int n) {
  for (int i = 0; i < 2; ++i) f();
  hipStreamSynchronize(mainS);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  hipStreamSynchronize(mainS);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  CK(hipStreamCreate(&mainS)); CK(hipStreamCreate(&sideS)); CK(hipStreamCreate(&side2S));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const char* names[] = {"main only", "side only", "forked (1 side chain)", "forked (2 side chains)"};
  for (int mode = 0; mode < 4; ++mode) {
    double eager = timeit([&] { body(mode); }, 5);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(mainS, hipStreamCaptureModeGlobal));
    body(mode);
    CK(hipStreamEndCapture(mainS, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double graph = timeit([&] { hipGraphLaunch(ge, mainS); }, 5);
    printf("%-24s eager %8.1f us   graph %8.1f us\n", names[mode], eager, graph);
  }
  return 0;
}
