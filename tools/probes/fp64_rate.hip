// Probe: sustained fp64 vector issue rate on gfx950 for v_fma_f64 / v_mul_f64+v_add_f64 (unfused) streams, as a function of
// waves per SIMD.  Prints TFLOP/s (an FMA = 2 flop, mul or add = 1 flop) and cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, int iters, double a, double b) {
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) x[i] = __builtin_fma(x[i], a, b);          // 1 instr, 2 flop
      else { x[i] = x[i] * a; x[i] = x[i] + b; }                // 2 instrs, 2 flop
    }
  }
  double s = 0; for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 1024 * 16);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++)
    for (int wps = 1; wps <= 8; wps *= 2) {             // waves per SIMD: block = 256 threads (4 waves = 1 per SIMD), blocks per CU = wps
      const int blocks = 256 * wps;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 256 * iters * 8 * 2;
      const double instr_per_wave = (double)iters * 8 * (mode == 0 ? 1 : 2);
      const double cyc = ms * 1e-3 * 2.4e9;              // nominal 2.4 GHz
      printf("%s  waves/SIMD %d: %.2f ms  %.1f TFLOP/s  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", mode == 0 ? "fma    " : "mul+add", wps, ms, flops / ms / 1e9, cyc / (instr_per_wave * wps));
    }
  return 0;
}
