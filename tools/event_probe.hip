// What an event record costs the stream it is recorded on (MI355X, ROCm 7.2): N back-to-back launches of a ~20-us kernel
//   (a) plain   (b) hipEventRecord after every launch, another stream waits for it   (c) the event attached to the launch itself
//   (hipExtLaunchKernelGGL stopEvent: the dispatch packet's own completion signal, no marker packet)   (d) as (b), nobody waits.
// build: hipcc --offload-arch=gfx950 -O2 tools/event_probe.hip -o build/event_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int n) {
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.f) p[0] = v;
}
__global__ void tiny(float* p) { if (p[0] == 123.f) p[1] = 0.f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int N = 200, SPIN = 12000;
  float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto run = [&](int mode) -> double {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, nullptr, ev[i], 0, d, SPIN);
      else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, d, SPIN);
      if (mode == 1 || mode == 3) hipEventRecord(ev[i], s0);
      if (mode == 1 || mode == 2) { hipStreamWaitEvent(s1, ev[i], 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d); }
      if (mode == 4) { hipStreamWaitEvent(s0, ev[(i + N - 1) % N], 0); }   // a wait for an event that has long fired
    }
    hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
  };
  const char* names[] = {"plain", "record + other stream waits", "event on the launch (hipExt) + other stream waits", "record, nobody waits", "wait for a fired event before each launch"};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 5; ++m) printf("%-55s %8.2f us per launch\n", names[m], run(m));
  return 0;
}
