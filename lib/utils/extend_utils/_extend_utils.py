"""Stand-in for the reference's cffi-built extension `lib.utils.extend_utils._extend_utils`
(lib/utils/extend_utils/build_extend_utils_cffi.py: Ceres + CUDA + OpenMP objects behind `ffi` / `lib`).

The reference's `extend_utils.py:3` does `from lib.utils.extend_utils._extend_utils import lib, ffi` and then only ever
uses `ffi.cast("<ctype> *", ndarray.ctypes.data)` and `lib.<function>(pointers..., ints...)`.  Both objects are provided
here on ctypes (cffi is not needed), bound to this repository's native libraries under the reference's own C symbols
(src/utils_python_binding.h):

    uncertainty_pnp                         libpvnet_pnp.so   (dense LM instead of Ceres, include/pvnet_pnp.h)
    farthest_point_sampling[_init_center]   libpvnet_pnp.so
    findNearestPointIdxLauncher             libpvnet_vote.so  (HIP brute-force nearest neighbour, include/pvnet_nn.h;
                                                              host pointers in and out as the reference's launcher; needs a GPU)
    mesh_binary_rasterization, anything else not provided (rendering / data synthesis: out of scope);
                                            delegated to the checkout's own cffi-built module when it has one

With this file ahead of the reference checkout on sys.path (the `lib` packages are namespace packages: the two trees
merge), the reference's own `lib/utils/extend_utils/extend_utils.py` and `lib/utils/evaluation_utils.py` run unchanged on
the native libraries (tests/test_reference_callers.py)."""
import ctypes as C
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


class _FFI:
    """the two cffi calls the reference makes"""
    NULL = None

    @staticmethod
    def cast(ctype, value):
        if not ctype.strip().endswith("*"):
            raise TypeError(f"ffi.cast: only pointer casts are supported by this stand-in, not {ctype!r}")
        return C.c_void_p(int(value))


class _Lib:
    def __init__(self):
        self._pnp = None
        self._vote = None

    def _pnp_lib(self):
        if self._pnp is None:
            from pvnet_amd import build
            path = build.PNP_LIB
            if not os.path.exists(path):
                raise RuntimeError(f"{path} is missing: python -m pvnet_amd.build")
            L = C.CDLL(path)
            L.uncertainty_pnp.restype = None
            L.uncertainty_pnp.argtypes = [C.c_void_p] * 6 + [C.c_int]
            for n in ("farthest_point_sampling", "farthest_point_sampling_init_center"):
                getattr(L, n).restype = None
                getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            self._pnp = L
        return self._pnp

    def _vote_lib(self):
        if self._vote is None:
            from pvnet_amd import voting
            L = voting.load_library()  # raises when the HIP library is missing: there is no CPU fallback
            L.findNearestPointIdxLauncher.restype = None
            L.findNearestPointIdxLauncher.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5
            self._vote = L
        return self._vote

    # -- src/utils_python_binding.h, same names and argument order ------------------------------------------------
    def uncertainty_pnp(self, pts2d, pts3d, wgt2d, K, init_rt, result_rt, pn):
        self._pnp_lib().uncertainty_pnp(pts2d, pts3d, wgt2d, K, init_rt, result_rt, int(pn))

    def farthest_point_sampling(self, pts, idxs, pn, sn):
        self._pnp_lib().farthest_point_sampling(pts, idxs, int(pn), int(sn))

    def farthest_point_sampling_init_center(self, pts, idxs, pn, sn):
        self._pnp_lib().farthest_point_sampling_init_center(pts, idxs, int(pn), int(sn))

    def findNearestPointIdxLauncher(self, ref_pts, que_pts, idxs, b, pn1, pn2, dim, exclude_self):
        self._vote_lib().findNearestPointIdxLauncher(ref_pts, que_pts, idxs, int(b), int(pn1), int(pn2), int(dim),
                                                     int(exclude_self))

    def mesh_binary_rasterization(self, *a):
        up = _upstream()
        if up is not None:  # the user's checkout has its own cffi-built module: rendering stays with it
            return up.lib.mesh_binary_rasterization(*a)
        raise NotImplementedError("mesh_binary_rasterization (rendering) is out of scope of this layer -- SURVEY.md section 2 "
                                  "-- and no cffi-built _extend_utils of a zju3dv/pvnet checkout was found on sys.path")

    def __getattr__(self, name):  # anything else the reference's header may declare (render_depth_cffi, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        up = _upstream()
        if up is not None and hasattr(up.lib, name):
            return getattr(up.lib, name)
        raise AttributeError(f"_extend_utils.lib.{name}: not provided by the MI355X stand-in and no cffi-built module found")


_up = False


def _upstream():
    """the reference's own cffi-built `_extend_utils` extension module, if a checkout further down sys.path has one"""
    global _up
    if _up is False:
        _up = None
        import importlib.machinery
        import importlib.util
        here = os.path.dirname(os.path.abspath(__file__))
        for root in sys.path:
            d = os.path.join(root or ".", "lib", "utils", "extend_utils")
            if not os.path.isdir(d) or os.path.abspath(d) == here:
                continue
            for suffix in importlib.machinery.EXTENSION_SUFFIXES:
                cand = os.path.join(d, "_extend_utils" + suffix)
                if os.path.exists(cand):
                    spec = importlib.util.spec_from_file_location("lib.utils.extend_utils._extend_utils_upstream", cand)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    _up = mod
                    return _up
    return _up


ffi = _FFI()
lib = _Lib()
