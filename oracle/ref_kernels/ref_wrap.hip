// oracle/_ref build unit (TEST INFRASTRUCTURE, never part of the product): compiles the reference's own CUDA
// kernels -- lib/ransac_voting_gpu_layer/src/ransac_voting_kernel.cu, included from where it lies under
// /root/reference at build time (path passed as PVNET_REF_CU) -- for gfx950, and exposes the two launchers of the
// voting path behind a C ABI on raw device pointers, so that the GPU parity tests can run the reference's device
// code on the MI355X next to ours.  Nothing of the reference is copied into this repository.
#include PVNET_REF_CU

extern "C" {
// ransac_voting.generate_hypothesis(direct [tn,vn,2], coords [tn,2], idxs [hn,vn,2]) -> [hn,vn,2]
int ref_generate_hypothesis(const float* direct, const float* coords, const int* idxs, float* hypo_pts, int tn, int vn,
                            int hn) {
    at::Tensor out = generate_hypothesis_launcher(at::Tensor::wrap(direct, tn, vn, 2), at::Tensor::wrap(coords, tn, 2, 1),
                                                  at::Tensor::wrap(idxs, hn, vn, 2));
    hipError_t e = hipMemcpy(hypo_pts, out.data<float>(), sizeof(float) * (size_t)hn * vn * 2, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return (int)e;
}
// ransac_voting.voting_for_hypothesis(direct, coords, hypo_pts [hn,vn,2], inliers [hn,vn,tn] uint8, thresh)
int ref_voting_for_hypothesis(const float* direct, const float* coords, const float* hypo_pts, unsigned char* inliers,
                              int tn, int vn, int hn, float inlier_thresh) {
    voting_for_hypothesis_launcher(at::Tensor::wrap(direct, tn, vn, 2), at::Tensor::wrap(coords, tn, 2, 1),
                                   at::Tensor::wrap(hypo_pts, hn, vn, 2), at::Tensor::wrap(inliers, hn, vn, tn),
                                   inlier_thresh);
    return (int)hipDeviceSynchronize();
}
// the vanishing-point pair (ransac_voting.cpp:57-99): hypo_pts [hn,vn,3]
int ref_generate_hypothesis_vanishing_point(const float* direct, const float* coords, const int* idxs, float* hypo_pts,
                                            int tn, int vn, int hn) {
    at::Tensor out = generate_hypothesis_vanishing_point_launcher(
        at::Tensor::wrap(direct, tn, vn, 2), at::Tensor::wrap(coords, tn, 2, 1), at::Tensor::wrap(idxs, hn, vn, 2));
    hipError_t e = hipMemcpy(hypo_pts, out.data<float>(), sizeof(float) * (size_t)hn * vn * 3, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return (int)e;
}
int ref_voting_for_hypothesis_vanishing_point(const float* direct, const float* coords, const float* hypo_pts,
                                              unsigned char* inliers, int tn, int vn, int hn, float inlier_thresh) {
    voting_for_hypothesis_vanishing_point_launcher(at::Tensor::wrap(direct, tn, vn, 2), at::Tensor::wrap(coords, tn, 2, 1),
                                                   at::Tensor::wrap(hypo_pts, hn, vn, 3),
                                                   at::Tensor::wrap(inliers, hn, vn, tn), inlier_thresh);
    return (int)hipDeviceSynchronize();
}
const char* ref_build_info(void) {
#ifdef PVNET_REF_CONTRACT
    return "reference kernels, gfx950, fp-contract=" PVNET_REF_CONTRACT;
#else
    return "reference kernels, gfx950";
#endif
}
}
