// Phase timing of row_down_kernel<20> (wall-clock stamps of the first/last thread of two workgroups, 100 MHz ticks = 10 ns).
#define GVK_STAMPS 1
#include "../../gaviko_amd/csrc/rowwise.hip"
#include <cstdio>
#include <vector>
namespace gvk { int set_error(int c, const char*, ...) { return c; } int check_launch(const char*) { return 0; } bool plan_recording() { return false; } void plan_push(std::function<void()>&&) {} }
int main() {
  using namespace gvk;
  const int M = 4132, C = 768, L = 20;
  float *x, *w, *b, *y, *z;
  hipMalloc(&x, (size_t)M * C * 4); hipMalloc(&w, L * C * 4); hipMalloc(&b, L * 4); hipMalloc(&y, M * L * 4); hipMalloc(&z, M * L * 4);
  hipMemset(x, 0, (size_t)M * C * 4); hipMemset(w, 0, L * C * 4); hipMemset(b, 0, L * 4);
  for (int layout = 0; layout < 2; ++layout) {
    DownArgs a{}; a.x = x; a.w = w; a.bias = b; a.y = y; a.z = z; a.M = M; a.C = C; a.act = 1; a.w_layout = layout; a.eps = 1e-5f;
    for (int it = 0; it < 3; ++it) launch_row_down(a, L, 0);
    hipDeviceSynchronize();
    long long st[4][16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    const char* nm[] = {"wg0 t0", "wgMid t0", "wg0 tLast", "wgMid tLast"};
    for (int r = 0; r < 4; ++r) {
      printf("layout %d %-12s", layout, nm[r]);
      for (int i = 1; i <= 6; ++i) printf("  ph%d %+6.2f us", i, (st[r][i] - st[r][i - 1]) * 0.01);
      printf("   total %.2f us (start offset vs wg0 %+.2f)\n", (st[r][6] - st[r][0]) * 0.01, (st[r][0] - st[0][0]) * 0.01);
    }
  }
  return 0;
}
