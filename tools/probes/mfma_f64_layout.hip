// Probe: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (used by the Schur-complement tiles of ba_device.h).
// Build: hipcc --offload-arch=gfx950 -O2 mfma_f64_layout.hip -o mfma_probe.  Every lane feeds a = A[ia(l)][ka(l)],
// b = B[kb(l)][jb(l)] for the candidate lane maps below and the raw accumulator registers are matched against A*B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* a_in, const double* b_in, double* raw) {
  const int l = threadIdx.x;
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a_in[l], b_in[l], c, 0, 0, 0);
  for (int v = 0; v < 4; v++) raw[l * 4 + v] = c[v];
}
int main() {
  double hA[64], hB[64], ref[256], raw[256], la[64], lb[64];
  for (int i = 0; i < 64; i++) { hA[i] = 1 + i * 0.5; hB[i] = 2 - i * 0.25 + (i % 7) * 0.125; }   // A[i][k] = hA[i*4+k], B[k][j] = hB[k*16+j]
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int kk = 0; kk < 4; kk++) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dR;
  hipMalloc(&dA, sizeof la); hipMalloc(&dB, sizeof lb); hipMalloc(&dR, sizeof raw);
  int found = 0;
  for (int am = 0; am < 2; am++) for (int bm = 0; bm < 2; bm++) {
    for (int l = 0; l < 64; l++) {
      const int ia = am ? l / 4 : l % 16, ka = am ? l % 4 : l / 16;
      const int jb = bm ? l / 4 : l % 16, kb = bm ? l % 4 : l / 16;
      la[l] = hA[ia * 4 + ka]; lb[l] = hB[kb * 16 + jb];
    }
    hipMemcpy(dA, la, sizeof la, hipMemcpyHostToDevice); hipMemcpy(dB, lb, sizeof lb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dR);
    hipMemcpy(raw, dR, sizeof raw, hipMemcpyDeviceToHost);
    for (int dm = 0; dm < 4; dm++) {
      double e = 0;
      for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
        int i, j;
        if (dm == 0) { i = 4 * (l / 16) + v; j = l % 16; }
        else if (dm == 1) { i = l % 16; j = 4 * (l / 16) + v; }
        else if (dm == 2) { i = (l / 16) + 4 * v; j = l % 16; }
        else { i = l % 16; j = (l / 16) + 4 * v; }
        double d = raw[l * 4 + v] - ref[i * 16 + j]; if (d < 0) d = -d; if (d > e) e = d;
      }
      if (e < 1e-9) { printf("MATCH: A map %d (0: A[l%%16][l/16], 1: A[l/4][l%%4]), B map %d (0: B[l/16][l%%16], 1: B[l%%4][l/4]), D map %d (0: D[4(l/16)+v][l%%16], 1: D[l%%16][4(l/16)+v], 2: D[l/16+4v][l%%16], 3: D[l%%16][l/16+4v])\n", am, bm, dm); found++; }
    }
  }
  if (!found) printf("no candidate layout matched\n");
  return found ? 0 : 1;
}
