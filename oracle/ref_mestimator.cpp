// Driver (ours) around the reference's jni/MEstimator.h, which is included from where it lies
// under the reference checkout and compiled verbatim.  Output: oracle/_ref/libref_mestimator.so.
// Test infrastructure only; used to pin the oracle's M-estimator restatement.
#include <cmath>   // the reference header uses sqrt/log without including <cmath>
#include REF_MESTIMATOR_H

template <class M>
static double sigma(const double* v, int n) { std::vector<double> x(v, v + n); return M::FindSigmaSquared(x); }

extern "C" {
// est: 0 Tukey, 1 Cauchy, 2 Huber, 3 LeastSquares
double ref_find_sigma_squared(int est, const double* v, int n) {
  switch (est) { case 0: return sigma<Tukey>(v, n); case 1: return sigma<Cauchy>(v, n);
                 case 2: return sigma<Huber>(v, n); default: return sigma<LeastSquares>(v, n); }
}
double ref_weight(int est, double e2, double s2) {
  switch (est) { case 0: return Tukey::Weight(e2, s2); case 1: return Cauchy::Weight(e2, s2);
                 case 2: return Huber::Weight(e2, s2); default: return LeastSquares::Weight(e2, s2); }
}
double ref_sqrt_weight(int est, double e2, double s2) {
  switch (est) { case 0: return Tukey::SquareRootWeight(e2, s2); case 1: return Cauchy::SquareRootWeight(e2, s2);
                 case 2: return Huber::SquareRootWeight(e2, s2); default: return LeastSquares::SquareRootWeight(e2, s2); }
}
double ref_objective(int est, double e2, double s2) {
  switch (est) { case 0: return Tukey::ObjectiveScore(e2, s2); case 1: return Cauchy::ObjectiveScore(e2, s2);
                 case 2: return Huber::ObjectiveScore(e2, s2); default: return LeastSquares::ObjectiveScore(e2, s2); }
}
}
