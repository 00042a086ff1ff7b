// Cross-stream hand-off cost on MI355X: ping-pong of short kernels over two streams with (a) hipEventRecord + hipStreamWaitEvent,
// (b) hipStreamWriteValue32 + hipStreamWaitValue32 (PLAIN=1: on hipMalloc memory, default: 8-byte signal memory).
// hipcc -O2 --offload-arch=gfx950 -o probe_handoff_smo probe_handoff_smo.hip; result: profiles/r04_probe_handoff_smo.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <cstdlib>
__global__ void spin(long ticks, int* sink) {
  const long t0 = clock64();
  while (clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  const int n = 400;
  for (long ticks : {2000L, 20000L}) {
    // single stream
    for (int i = 0; i < 10; ++i) spin<<<1, 64, 0, A>>>(ticks, nullptr);
    CK(hipStreamSynchronize(A));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) spin<<<1, 64, 0, A>>>(ticks, nullptr);
    CK(hipStreamSynchronize(A));
    double single = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    // events
    hipEvent_t ev[2 * 400];
    for (auto& evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
      hipStream_t s = (i & 1) ? B : A, o = (i & 1) ? A : B;
      spin<<<1, 64, 0, s>>>(ticks, nullptr);
      CK(hipEventRecord(ev[i], s));
      CK(hipStreamWaitEvent(o, ev[i], 0));
    }
    CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
    double evt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    // stream memory operations
    int* flag = nullptr;
    hipError_t e = hipSuccess;
    int* base = nullptr;
    if (getenv("PLAIN")) { CK(hipMalloc(&base, 4096)); flag = base + 37; }
    else {
      e = hipExtMallocWithFlags((void**)&flag, getenv("SIGBYTES") ? atoi(getenv("SIGBYTES")) : 8, hipMallocSignalMemory);
      if (e != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(e)); return 1; }
      if (getenv("SIGBYTES")) flag += 5;
    }
    CK(hipMemset(flag, 0, 4));
    double smo = -1;
    t0 = std::chrono::steady_clock::now();
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
      hipStream_t s = (i & 1) ? B : A, o = (i & 1) ? A : B;
      spin<<<1, 64, 0, s>>>(ticks, nullptr);
      e = hipStreamWriteValue32(s, flag, i + 1, 0);
      if (e != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); ok = false; break; }
      e = hipStreamWaitValue32(o, flag, i + 1, hipStreamWaitValueGte, 0xffffffffu);
      if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); ok = false; break; }
    }
    CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
    if (ok) smo = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("spin %ld ticks: single stream %.2f us/kernel; ping-pong with events %.2f (+%.2f); with stream memory ops %.2f (+%.2f)\n",
           ticks, single, evt, evt - single, smo, smo - single);
  }
  return 0;
}
