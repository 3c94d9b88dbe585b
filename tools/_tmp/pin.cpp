#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main() {
  const size_t n = 64u << 20;
  char* h = (char*)malloc(n); memset(h, 1, n);
  for (unsigned flags : {0u, 0x20000000u /*NumaUser*/, 0x80000000u /*NonCoherent*/, 0x1u /*Portable*/}) {
    char* p = nullptr;
    if (hipHostMalloc((void**)&p, n, flags) != hipSuccess) { printf("flags %x failed\n", flags); continue; }
    memset(p, 1, n);
    auto t = std::chrono::steady_clock::now(); memset(p, 2, n); double w = ms(t);
    t = std::chrono::steady_clock::now(); volatile long s = 0; for (size_t i = 0; i < n; i += 64) s += p[i]; double r = ms(t);
    t = std::chrono::steady_clock::now(); memcpy(p, h, n); double c = ms(t);
    printf("hipHostMalloc flags %08x: memset %.2f ms (%.1f GB/s)  strided read %.2f ms  memcpy from heap %.2f ms\n", flags, w, n / w / 1e6, r, c);
    hipHostFree(p);
  }
  auto t = std::chrono::steady_clock::now(); memset(h, 2, n); double w = ms(t);
  t = std::chrono::steady_clock::now(); volatile long s = 0; for (size_t i = 0; i < n; i += 64) s += h[i]; double r = ms(t);
  printf("malloc: memset %.2f ms (%.1f GB/s) strided read %.2f ms\n", w, n / w / 1e6, r);
  // registered heap memory
  if (hipHostRegister(h, n, hipHostRegisterDefault) == hipSuccess) {
    t = std::chrono::steady_clock::now(); memset(h, 3, n); w = ms(t);
    printf("hipHostRegister'ed malloc: memset %.2f ms (%.1f GB/s)\n", w, n / w / 1e6);
    void* d = nullptr; hipMalloc(&d, n);
    hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    t = std::chrono::steady_clock::now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); printf("H2D from registered: %.2f ms (%.1f GB/s)\n", ms(t), n / ms(t) / 1e6);
    hipHostUnregister(h);
  }
  return 0;
}
