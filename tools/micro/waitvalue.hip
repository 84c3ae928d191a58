// micro-benchmark: what does a satisfied hipStreamWaitValue32 between two kernels cost on the stream? (profiles/r03_split_forms.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void work(double *p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
__global__ void setflag(unsigned *f, unsigned v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0)); printf("CanUseStreamWaitValue %d\n", can);
  const int n = 1 << 22; double *p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8));
  unsigned *flag = nullptr, *sig = nullptr;
  CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  if (hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory) != hipSuccess) { printf("no signal memory\n"); sig = nullptr; (void)hipGetLastError(); }
  else CK(hipMemset(sig, 0, 8));
  hipStream_t s, c; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
  const int iters = argc > 2 ? atoi(argv[2]) : 500;
  for (int mode = 0; mode < 5; mode++) {
    if (only >= 0 && mode != only) continue;
    printf("mode %d starts\n", mode);
    unsigned *f = (mode == 2 || mode == 4) ? sig : flag;
    if ((mode == 2 || mode == 4) && !sig) continue;
    if (f) { hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, s, f, 1u); }
    CK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 2; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; i++) {
        hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, s, p, n);
        if (mode == 1 || mode == 2) CK(hipStreamWaitValue32(s, f, 1u, hipStreamWaitValueGte, 0xffffffffu));
        if (mode == 3 || mode == 4) { // the flag is produced by the other stream each iteration: c sets i+2 ... s waits for it
          hipLaunchKernelGGL(setflag, dim3(1), dim3(1), 0, c, f, (unsigned)(i + 2));
          CK(hipStreamWaitValue32(s, f, (unsigned)(i + 2), hipStreamWaitValueGte, 0xffffffffu));
        }
      }
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(c));
      auto t1 = std::chrono::steady_clock::now();
      if (rep) printf("mode %d (%s): %.2f us per iteration\n", mode,
                      mode == 0 ? "kernel only" : mode == 1 ? "kernel + satisfied wait, device memory" : mode == 2 ? "kernel + satisfied wait, signal memory" :
                      mode == 3 ? "kernel + wait for a flag the other stream sets, device memory" : "same, signal memory",
                      std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
    }
    if (f) { CK(hipMemset(f, 0, 8)); }
  }
  return 0;
}
