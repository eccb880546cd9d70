// oracle/_ref build unit (TEST INFRASTRUCTURE, never part of the product): compiles the reference's own nearest-
// neighbour kernels and host launcher -- lib/utils/extend_utils/src/nearest_neighborhood.cu, included from where it lies
// under /root/reference at build time (path passed as PVNET_REF_NN_CU) -- for gfx950.  The file already exports its
// launcher `findNearestPointIdxLauncher` with C linkage (host pointers in and out; it allocates, copies and frees device
// memory itself), so nothing is wrapped: the GPU parity test calls the reference's own entry point on the MI355X next to
// the product's.  Nothing of the reference is copied into this repository.
#include PVNET_REF_NN_CU

extern "C" const char* ref_nn_build_info(void) {
#ifdef PVNET_REF_CONTRACT
    return "reference nearest_neighborhood.cu, gfx950, fp-contract=" PVNET_REF_CONTRACT;
#else
    return "reference nearest_neighborhood.cu, gfx950";
#endif
}
