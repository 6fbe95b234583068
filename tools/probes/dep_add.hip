// Probe: cycles per DEPENDENT v_add_f64 / v_fma_f64 / v_mul_f64 of one wavefront, alone on its SIMD and beside a second wavefront
// that issues independent fp64 work.  hipcc --offload-arch=gfx950 -O3 dep_add.hip -o dep_add && ./dep_add
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned long long* cyc, int n, int mode) {
  const int wave = threadIdx.x >> 6;
  double a = out[threadIdx.x], b = out[threadIdx.x + 256], c0 = a, c1 = b, c2 = a + 1, c3 = b + 1;
  __syncthreads();
  const unsigned long long t0 = clock64();
  if (wave == 0) {
    for (int i = 0; i < n; i++) {     // 8 dependent additions per trip
      asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                   "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    }
  } else if (mode == 1) {
    for (int i = 0; i < n; i++) {     // independent multiplies: a full-rate neighbour
      asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                   "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b));
    }
  } else if (mode == 2) {
    for (int i = 0; i < n; i++) {     // dependent chain in the neighbour too
      asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n"
                   "v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1" : "+v"(c0) : "v"(b));
    }
  }
  const unsigned long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  out[threadIdx.x] = a + c0 + c1 + c2 + c3;
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 4096 * 8); hipMalloc(&c, 64 * 8); hipMemset(d, 0, 4096 * 8);
  const int n = 20000;
  // waves of a workgroup land on different SIMDs: 64-thread blocks of mode 0 = alone; 5 waves (320 threads): wave 4 shares SIMD 0 with wave 0
  for (int threads : {64, 320}) for (int mode : {0, 1, 2}) {
    if (threads == 64 && mode) continue;
    k<<<1, threads>>>(d, c, n, mode); hipDeviceSynchronize();
    k<<<1, threads>>>(d, c, n, mode); hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %d mode %d: wave0 %.2f cycles per dependent add", threads, mode, (double)h[0] / (8.0 * n));
    for (int w = 1; w < threads / 64; w++) printf("  wave%d %.2f/instr", w, (double)h[w] / (8.0 * n));
    printf("\n");
  }
  return 0;
}
