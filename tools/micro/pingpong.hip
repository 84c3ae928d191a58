// micro-benchmark: wake-up latency of a cross-stream dependency -- stream A runs a kernel and signals, stream B waits and runs a kernel and
// signals back -- for the three ways to wait: hipStreamWaitValue32 (command processor), a one-wave polling kernel, hipStreamWaitEvent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned *f, unsigned v) {   // ~ticks of the 100 MHz clock, then publish v
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (f) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void waitk(const unsigned *f, unsigned v) { while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) __builtin_amdgcn_s_sleep(8); }
int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = 300;
  const unsigned long long work = argc > 2 ? atoll(argv[2]) : 1000;   // 10 us
  unsigned *fa, *fb; CK(hipMalloc(&fa, 64)); CK(hipMalloc(&fb, 64)); CK(hipMemset(fa, 0, 64)); CK(hipMemset(fb, 0, 64)); CK(hipDeviceSynchronize());
  hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  hipEvent_t ea[512], eb[512];
  for (int i = 0; i < 512; i++) { CK(hipEventCreateWithFlags(&ea[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming)); }
  for (int rep = 0; rep < 2; rep++) {
    const unsigned base = rep * 1000 + 1;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++) {
      const unsigned v = base + i;
      // A: work, signal fa = v ;  B: wait fa >= v, work, signal fb = v ;  A: wait fb >= v (next iteration starts behind it)
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, A, work, fa, v);
      if (mode == 2) CK(hipEventRecord(ea[i], A));
      if (mode == 0) CK(hipStreamWaitValue32(B, fa, v, hipStreamWaitValueGte, 0xffffffffu));
      else if (mode == 1) hipLaunchKernelGGL(waitk, dim3(1), dim3(64), 0, B, fa, v);
      else CK(hipStreamWaitEvent(B, ea[i], 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, B, work, fb, v);
      if (mode == 2) CK(hipEventRecord(eb[i], B));
      if (mode == 0) CK(hipStreamWaitValue32(A, fb, v, hipStreamWaitValueGte, 0xffffffffu));
      else if (mode == 1) hipLaunchKernelGGL(waitk, dim3(1), dim3(64), 0, A, fb, v);
      else CK(hipStreamWaitEvent(A, eb[i], 0));
    }
    CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
    auto t1 = std::chrono::steady_clock::now();
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
    if (rep) printf("mode %d (%s), work %.1f us: %.2f us per round trip = 2 x (work + hand-off): hand-off %.2f us\n", mode,
                    mode == 0 ? "hipStreamWaitValue32" : mode == 1 ? "polling kernel" : "hipStreamWaitEvent", work / 100.0, us, us / 2 - work / 100.0);
  }
  return 0;
}
