// Test infrastructure (oracle/): the reference's uncertainty-PnP cost functor, compiled FROM THE REFERENCE TREE
// (PVNET_REF_PNP_CPP = lib/utils/extend_utils/src/uncertainty_pnp.cpp, included where it lies; nothing is copied) against
// the reference's vendored ceres/jet.h + ceres/rotation.h.  Two entry points:
//   ref_pnp_residuals: ReprojectionErrorArray::operator()<double>            (uncertainty_pnp.cpp:16-35)
//   ref_pnp_jacobian:  the same operator on ceres::Jet<double, 6> with pose[k] = (value, e_k) -- what
//                      ceres::AutoDiffCostFunction<ReprojectionErrorArray, 2, 6> (:46-47) hands to the solver.
// Only tests/ load the resulting oracle/_ref/libpvnet_refpnp.so.
#include PVNET_REF_PNP_CPP

extern "C" {
const char* ref_pnp_build_info() { return "reference ReprojectionErrorArray (uncertainty_pnp.cpp) on vendored ceres jet.h/rotation.h"; }

// pts2d [pn,2], pts3d [pn,3], wgt2d [pn,3], K [3,3], pose [6] -> residuals [2 pn]
void ref_pnp_residuals(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K, const double* pose,
                       int pn, double* residuals) {
    for (int i = 0; i < pn; ++i) {
        ReprojectionErrorArray f(pts2d[i * 2], pts2d[i * 2 + 1], pts3d[i * 3], pts3d[i * 3 + 1], pts3d[i * 3 + 2],
                                 wgt2d[i * 3 + 0], wgt2d[i * 3 + 1], wgt2d[i * 3 + 2], K[0], K[4], K[2], K[5]);
        f(pose, residuals + 2 * i);
    }
}

// ... -> residuals [2 pn], jacobian [2 pn][6]
void ref_pnp_jacobian(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K, const double* pose,
                      int pn, double* residuals, double* jacobian) {
    typedef ceres::Jet<double, 6> J6;
    J6 p[6];
    for (int k = 0; k < 6; ++k) p[k] = J6(pose[k], k);
    for (int i = 0; i < pn; ++i) {
        ReprojectionErrorArray f(pts2d[i * 2], pts2d[i * 2 + 1], pts3d[i * 3], pts3d[i * 3 + 1], pts3d[i * 3 + 2],
                                 wgt2d[i * 3 + 0], wgt2d[i * 3 + 1], wgt2d[i * 3 + 2], K[0], K[4], K[2], K[5]);
        J6 r[2];
        f(p, r);
        for (int c = 0; c < 2; ++c) {
            residuals[2 * i + c] = r[c].a;
            for (int k = 0; k < 6; ++k) jacobian[(2 * i + c) * 6 + k] = r[c].v[k];
        }
    }
}
}
