/* Build shim (test infrastructure, oracle/_ref only): the few at::Tensor members the reference's launchers touch
 * (size, data<T>, type, at::zeros), backed by plain device pointers so that the launchers and kernels of
 * ransac_voting_kernel.cu run unmodified without libtorch. */
#ifndef PVNET_REF_SHIM_ATEN_H
#define PVNET_REF_SHIM_ATEN_H
#include <hip/hip_runtime.h>
#include <initializer_list>
#include <memory>
namespace at {
struct Type {};
struct Storage {
    void* p = nullptr;
    bool owned = false;
    ~Storage() { if (owned && p) (void)hipFree(p); }
};
struct Tensor {
    std::shared_ptr<Storage> st;
    long dims[4] = {1, 1, 1, 1};
    long size(int i) const { return dims[i]; }
    template <typename T> T* data() const { return static_cast<T*>(st->p); }
    Type type() const { return Type(); }
    static Tensor wrap(const void* p, long d0, long d1, long d2) {
        Tensor t;
        t.st = std::make_shared<Storage>();
        t.st->p = const_cast<void*>(p);
        t.dims[0] = d0; t.dims[1] = d1; t.dims[2] = d2;
        return t;
    }
};
inline Tensor zeros(std::initializer_list<int> shape, Type) {  /* float32, zero-filled, on the current device */
    Tensor t;
    t.st = std::make_shared<Storage>();
    size_t n = 1;
    int i = 0;
    for (int d : shape) { t.dims[i++] = d; n *= (size_t)d; }
    if (hipMalloc(&t.st->p, n * sizeof(float) + 16) != hipSuccess) abort();
    t.st->owned = true;
    (void)hipMemset(t.st->p, 0, n * sizeof(float));
    return t;
}
}  // namespace at
#endif
