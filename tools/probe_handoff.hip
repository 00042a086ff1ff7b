// Cost of a cross-stream hand-off (hipEventRecord on one stream + hipStreamWaitEvent on another) between short kernels,
// against the same kernels back to back on one stream.  hipcc --offload-arch=gfx950 -O3 tools/probe_handoff.hip -o tools/probe_handoff
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long cycles, int* sink) {
  const long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}

int main() {
  hipStream_t P, Q;
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithFlags(&P, hipStreamNonBlocking);
  hipStreamCreateWithPriority(&Q, hipStreamNonBlocking, hi);
  const int n = 2000;
  std::vector<hipEvent_t> ev(2 * n);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (long cyc : {2000L, 10000L, 50000L}) {  // ~1, 4, 20 us at the 100 MHz-ish clock64? (s_memtime counts at 100 MHz on gfx9: adjust below)
    for (int rep = 0; rep < 2; ++rep) {
      // single stream: 2n kernels back to back
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 2 * n; ++i) spin<<<1, 64, 0, P>>>(cyc, nullptr);
      hipStreamSynchronize(P);
      auto t1 = std::chrono::steady_clock::now();
      // ping-pong: kernel on P, hand-off to Q, kernel on Q, hand-off to P
      for (int i = 0; i < n; ++i) {
        spin<<<1, 64, 0, P>>>(cyc, nullptr);
        hipEventRecord(ev[2 * i], P);
        hipStreamWaitEvent(Q, ev[2 * i], 0);
        spin<<<1, 64, 0, Q>>>(cyc, nullptr);
        hipEventRecord(ev[2 * i + 1], Q);
        hipStreamWaitEvent(P, ev[2 * i + 1], 0);
      }
      hipStreamSynchronize(P);
      hipStreamSynchronize(Q);
      auto t2 = std::chrono::steady_clock::now();
      const double single = std::chrono::duration<double, std::micro>(t1 - t0).count() / (2 * n);
      const double pingpong = std::chrono::duration<double, std::micro>(t2 - t1).count() / (2 * n);
      if (rep == 1)
        printf("spin %6ld ticks: single stream %.2f us per kernel, ping-pong over two streams %.2f us per kernel -> hand-off adds %.2f us\n",
               cyc, single, pingpong, pingpong - single);
    }
  }
  return 0;
}
