// ransac_voting -- the reference's compiled extension module on the MI355X library.
//
// Mirrors lib/ransac_voting_gpu_layer/src/ransac_voting.cpp:1-107: the same four pybind functions, argument lists,
// CHECK_INPUT behaviour (CUDA + contiguous, else an exception) and in/out conventions, bound to the C ABI of
// libpvnet_vote.so (include/pvnet_vote.h) instead of the CUDA launchers of ransac_voting_kernel.cu.  Built in-tree by
// pvnet_amd/build.py (g++ against the torch headers; no device code here) into
// lib/ransac_voting_gpu_layer/ransac_voting*.so, where the reference's driver imports it from
// (`import lib.ransac_voting_gpu_layer.ransac_voting as ransac_voting`, ransac_voting_gpu.py:2); an extension module
// takes precedence over the pure-Python stand-in of the same name next to it, which remains as the fallback.
//
// Differences from the reference binding, on purpose: the dtype is checked (the reference's `.data<float>()` would
// throw a less readable error), a failed launch raises instead of calling exit() (cuda_common.h gpuErrchk), and the
// kernels run on the CURRENT stream of the tensor's device, not on the legacy default stream.
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include "pvnet_vote.h"

#define CHECK_CUDA(x) TORCH_CHECK(x.is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) \
    CHECK_CUDA(x);     \
    CHECK_CONTIGUOUS(x)

namespace {

void* current_stream(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

struct Dims {
    int tn, vn, hn;
};

Dims check_pair(const at::Tensor& direct, const at::Tensor& coords, const at::Tensor& third, int64_t last, const char* what) {
    TORCH_CHECK(direct.scalar_type() == at::kFloat && coords.scalar_type() == at::kFloat, what, ": direct and coords must be float32");
    TORCH_CHECK(direct.dim() == 3 && direct.size(2) == 2, what, ": direct must be [tn,vn,2]");
    TORCH_CHECK(coords.dim() == 2 && coords.size(0) == direct.size(0) && coords.size(1) == 2, what, ": coords must be [tn,2]");
    TORCH_CHECK(third.dim() == 3 && third.size(1) == direct.size(1) && third.size(2) == last, what, ": third tensor must be [hn,vn,",
                last, "]");
    TORCH_CHECK(direct.size(0) > 0 && direct.size(1) > 0 && third.size(0) > 0, what, ": empty input");
    return {(int)direct.size(0), (int)direct.size(1), (int)third.size(0)};
}

}  // namespace

// ransac_voting.cpp:20-31
at::Tensor generate_hypothesis(at::Tensor direct, at::Tensor coords, at::Tensor idxs) {
    CHECK_INPUT(direct);
    CHECK_INPUT(coords);
    CHECK_INPUT(idxs);
    TORCH_CHECK(idxs.scalar_type() == at::kInt, "generate_hypothesis: idxs must be int32");
    const Dims d = check_pair(direct, coords, idxs, 2, "generate_hypothesis");
    c10::DeviceGuard guard(direct.device());  // (torch-ROCm presents its devices as 'cuda': the generic guard)
    at::Tensor hypo_pts = at::empty({d.hn, d.vn, 2}, direct.options());
    const int rc = pvnet_generate_hypothesis(direct.data_ptr<float>(), coords.data_ptr<float>(), idxs.data_ptr<int32_t>(),
                                             hypo_pts.data_ptr<float>(), d.tn, d.vn, d.hn, current_stream(direct));
    TORCH_CHECK(rc == 0, "pvnet_generate_hypothesis failed: ", rc);
    return hypo_pts;
}

// ransac_voting.cpp:41-55: in place, only ever sets ones
void voting_for_hypothesis(at::Tensor direct, at::Tensor coords, at::Tensor hypo_pts, at::Tensor inliers, float inlier_thresh) {
    CHECK_INPUT(direct);
    CHECK_INPUT(coords);
    CHECK_INPUT(hypo_pts);
    CHECK_INPUT(inliers);
    TORCH_CHECK(hypo_pts.scalar_type() == at::kFloat && inliers.scalar_type() == at::kByte,
                "voting_for_hypothesis: hypo_pts must be float32, inliers uint8");
    const Dims d = check_pair(direct, coords, hypo_pts, 2, "voting_for_hypothesis");
    TORCH_CHECK(inliers.dim() == 3 && inliers.size(0) == d.hn && inliers.size(1) == d.vn && inliers.size(2) == d.tn,
                "voting_for_hypothesis: inliers must be [hn,vn,tn]");
    c10::DeviceGuard guard(direct.device());  // (torch-ROCm presents its devices as 'cuda': the generic guard)
    const int rc = pvnet_voting_for_hypothesis(direct.data_ptr<float>(), coords.data_ptr<float>(), hypo_pts.data_ptr<float>(),
                                               inliers.data_ptr<uint8_t>(), d.tn, d.vn, d.hn, inlier_thresh, current_stream(direct));
    TORCH_CHECK(rc == 0, "pvnet_voting_for_hypothesis failed: ", rc);
}

// ransac_voting.cpp:64-75
at::Tensor generate_hypothesis_vanishing_point(at::Tensor direct, at::Tensor coords, at::Tensor idxs) {
    CHECK_INPUT(direct);
    CHECK_INPUT(coords);
    CHECK_INPUT(idxs);
    TORCH_CHECK(idxs.scalar_type() == at::kInt, "generate_hypothesis_vanishing_point: idxs must be int32");
    const Dims d = check_pair(direct, coords, idxs, 2, "generate_hypothesis_vanishing_point");
    c10::DeviceGuard guard(direct.device());  // (torch-ROCm presents its devices as 'cuda': the generic guard)
    at::Tensor hypo_pts = at::empty({d.hn, d.vn, 3}, direct.options());
    const int rc = pvnet_generate_hypothesis_vanishing_point(direct.data_ptr<float>(), coords.data_ptr<float>(),
                                                             idxs.data_ptr<int32_t>(), hypo_pts.data_ptr<float>(), d.tn, d.vn, d.hn,
                                                             current_stream(direct));
    TORCH_CHECK(rc == 0, "pvnet_generate_hypothesis_vanishing_point failed: ", rc);
    return hypo_pts;
}

// ransac_voting.cpp:85-99
void voting_for_hypothesis_vanishing_point(at::Tensor direct, at::Tensor coords, at::Tensor hypo_pts, at::Tensor inliers,
                                           float inlier_thresh) {
    CHECK_INPUT(direct);
    CHECK_INPUT(coords);
    CHECK_INPUT(hypo_pts);
    CHECK_INPUT(inliers);
    TORCH_CHECK(hypo_pts.scalar_type() == at::kFloat && inliers.scalar_type() == at::kByte,
                "voting_for_hypothesis_vanishing_point: hypo_pts must be float32, inliers uint8");
    const Dims d = check_pair(direct, coords, hypo_pts, 3, "voting_for_hypothesis_vanishing_point");
    TORCH_CHECK(inliers.dim() == 3 && inliers.size(0) == d.hn && inliers.size(1) == d.vn && inliers.size(2) == d.tn,
                "voting_for_hypothesis_vanishing_point: inliers must be [hn,vn,tn]");
    c10::DeviceGuard guard(direct.device());  // (torch-ROCm presents its devices as 'cuda': the generic guard)
    const int rc = pvnet_voting_for_hypothesis_vanishing_point(direct.data_ptr<float>(), coords.data_ptr<float>(),
                                                               hypo_pts.data_ptr<float>(), inliers.data_ptr<uint8_t>(), d.tn, d.vn,
                                                               d.hn, inlier_thresh, current_stream(direct));
    TORCH_CHECK(rc == 0, "pvnet_voting_for_hypothesis_vanishing_point failed: ", rc);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {  // ransac_voting.cpp:102-107, same names and doc strings
    m.def("generate_hypothesis", &generate_hypothesis, "generate hypothesis");
    m.def("voting_for_hypothesis", &voting_for_hypothesis, "voting for hypothesis");
    m.def("generate_hypothesis_vanishing_point", &generate_hypothesis_vanishing_point, "generate hypothesis vanishing point");
    m.def("voting_for_hypothesis_vanishing_point", &voting_for_hypothesis_vanishing_point, "voting for hypothesis vanishing point");
    m.attr("backend") = "pvnet_amd (libpvnet_vote.so, gfx950)";
}
