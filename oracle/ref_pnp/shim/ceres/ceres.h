// Test infrastructure (oracle/): a stand-in for <ceres/ceres.h> that lets the reference's
// lib/utils/extend_utils/src/uncertainty_pnp.cpp compile WITHOUT libceres, so that its cost functor
// (ReprojectionErrorArray::operator(), :16-35) can be evaluated -- with doubles and with ceres::Jet, i.e. exactly what
// ceres::AutoDiffCostFunction<ReprojectionErrorArray, 2, 6> evaluates (:46-47).  The functor's arithmetic comes from the
// reference's vendored, header-only ceres/jet.h and ceres/rotation.h (this directory precedes
// lib/utils/extend_utils/include on the include path and provides ONLY this one header); the solver classes the rest of
// the file names are empty shells: the vendored libceres.so.1.14.0 cannot be linked here (spqr, cholmod, cxsparse,
// openblas are absent), and nothing of the solver is called by the checker.
#ifndef PVNET_ORACLE_CERES_SHIM_H_
#define PVNET_ORACLE_CERES_SHIM_H_
#include <cstring>
#include "ceres/jet.h"
namespace ceres {
class CostFunction {
 public:
    virtual ~CostFunction() {}
};
template <typename Functor, int kNumResiduals, int N0>
class AutoDiffCostFunction : public CostFunction {
 public:
    explicit AutoDiffCostFunction(Functor* f) : functor(f) {}
    ~AutoDiffCostFunction() override { delete functor; }
    Functor* functor;
};
struct Problem {
    void AddResidualBlock(CostFunction* c, void*, double*) { delete c; }
};
enum LinearSolverType { DENSE_SCHUR };
struct Solver {
    struct Options { LinearSolverType linear_solver_type; bool minimizer_progress_to_stdout; };
    struct Summary {};
};
inline void Solve(const Solver::Options&, Problem*, Solver::Summary*) {}  // never called by the checker
}  // namespace ceres
#endif
