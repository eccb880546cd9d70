/* pvnet_pnp.h -- C ABI of the host-side pose refinement (libpvnet_pnp.so, plain C++, no GPU, no dependencies).
 *
 * the reference solves the 9-point pose problem on the host too --
 *   - `uncertainty_pnp`  (lib/utils/extend_utils/src/uncertainty_pnp.cpp:61-92, bound through cffi at
 *                         lib/utils/extend_utils/extend_utils.py:63-114): Ceres Levenberg-Marquardt (DENSE_SCHUR,
 *                         default options) on the 2x2-weighted reprojection residuals of uncertainty_pnp.cpp:18-35,
 *                         pose = (angle-axis[3], translation[3]);
 *   - `pnp`              (lib/utils/evaluation_utils.py:19-52): cv2.solvePnP(..., SOLVEPNP_ITERATIVE), i.e. the same
 *                         Levenberg-Marquardt on unweighted residuals after a linear start.
 * Neither Ceres nor OpenCV exists in this image, and the problem is 2*pn residuals x 6 parameters: a dense LM with
 * analytic Jacobians in ~200 lines.  `uncertainty_pnp` below keeps the reference's name, argument order and meaning
 * (uncertainty_pnp.cpp:61-69), so the reference's cffi `lib.uncertainty_pnp(...)` call binds to it unchanged.
 */
#ifndef PVNET_PNP_H_
#define PVNET_PNP_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Reference signature (uncertainty_pnp.cpp:61-69).  All pointers are host pointers to contiguous doubles.
 *   pts2d [pn,2], pts3d [pn,3], wgt2d [pn,3] = (wxx, wxy, wyy) of the symmetric 2x2 weight per point,
 *   K [3,3] row-major (fx = K[0], fy = K[4], px = K[2], py = K[5]), init_rt [6] = angle-axis + translation,
 *   result_rt [6] receives the refined pose.  Like the reference it returns nothing; a failed solve leaves the best
 *   iterate (at worst init_rt) in result_rt. */
void uncertainty_pnp(double* pts2d, double* pts3d, double* wgt2d, double* K, double* init_rt, double* result_rt,
                     int pn);

/* The cost function the solver minimises, exposed so that it can be pinned to the reference's: residuals [2 pn] =
 * (wxx dx + wxy dy, wxy dx + wyy dy) per point with (dx, dy) = projection - observation, exactly
 * ReprojectionErrorArray::operator() (uncertainty_pnp.cpp:16-35), and -- when `jacobian` is not NULL -- its analytic
 * Jacobian [2 pn][6] with respect to (angle-axis, translation), which is what ceres::AutoDiffCostFunction derives from
 * that functor by Jets (uncertainty_pnp.cpp:46-47).  wgt2d may be NULL (identity).  Returns 0, 1 if a point falls on the
 * camera plane (outputs undefined), -1 on bad arguments.  tests/test_pnp.py compares both with the reference's functor
 * compiled from the reference tree against its vendored ceres/jet.h + rotation.h (oracle/_ref/libpvnet_refpnp.so). */
int pvnet_pnp_evaluate(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K, const double* rt,
                       int pn, double* residuals, double* jacobian);

/* The same solver with a report.  wgt2d may be NULL (identity weights = the unweighted reprojection error that
 * cv2.solvePnP's ITERATIVE flag minimises).  Returns the number of LM iterations taken (>= 0), or -1 on bad
 * arguments.  final_cost (may be NULL) receives 0.5 * sum of squared weighted residuals. */
int pvnet_pnp_refine(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K,
                     const double* init_rt, double* result_rt, int pn, int max_iterations, double* final_cost);

/* The whole of `pnp` (lib/utils/evaluation_utils.py:19-52) / `uncertainty_pnp` (extend_utils.py:63-114) for one image:
 * linear start (DLT on normalised image points, projected onto SO(3); pn >= 6) + LM on the reprojection error, then --
 * if wgt2d is given -- LM on the 2x2-weighted residuals from there.  result_rt [6] = angle-axis + translation.
 * Returns the LM iterations taken (>= 0), -1 on bad arguments, -2 if the linear start is degenerate. */
int pvnet_pnp_solve(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K, double* result_rt,
                    int pn);
/* n images that share the object points: pts2d [n,pn,2], wgt2d [n,pn,3] or NULL, result_rt [n,6] (zeros where the
 * solve failed).  Returns the number of failed images, -1 on bad arguments. */
int pvnet_pnp_solve_batch(const double* pts2d, const double* pts3d, const double* wgt2d, const double* K,
                          double* result_rt, int n, int pn);
/* result_rt [n,6] of pvnet_pnp_solve_batch -> poses [n,3,4] = (R | t) row-major (zeros for a failed image): what `pnp` returns
 * (lib/utils/evaluation_utils.py:50-52), for a whole batch without a Python loop. */
void pvnet_pnp_poses_from_rt(const double* result_rt, double* poses, int n);

/* angle-axis <-> rotation matrix (cv2.Rodrigues / ceres::AngleAxisToRotationMatrix), row-major R[9] */
void pvnet_angle_axis_to_matrix(const double* aa, double* R);
void pvnet_matrix_to_angle_axis(const double* R, double* aa);

/* Farthest-point sampling under the reference's own C symbols (src/utils_python_binding.h, implemented in
 * src/farthest_point_sampling.cpp:178-221; called through cffi by extend_utils.py:22-37 -- the reference chooses its 8 object
 * key-points with it, lib/utils/data_utils.py:144).  pts [pn,3] float32, idxs [sn] int32 receives the selected indices in
 * selection order.  `_init_center` is deterministic (first point = farthest from the bounding-box centre, whose distances
 * also seed every point's nearest-selected distance); the plain one starts at a random point.  Float32 squared distances in
 * the reference's operation order; strict `>` from index 0 on ties. */
void farthest_point_sampling(float* pts, int* idxs, int pn, int sn);
void farthest_point_sampling_init_center(float* pts, int* idxs, int pn, int sn);

#ifdef __cplusplus
}
#endif
#endif /* PVNET_PNP_H_ */
