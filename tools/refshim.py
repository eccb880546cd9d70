"""sys.modules shims that let the reference's OWN scripts (tools/demo.py, tools/train_linemod.py of zju3dv/pvnet) be
imported and executed UNCHANGED in an image that lacks their third-party dependencies:
cv2, torchvision, easydict, plyfile, skimage, tensorboardX, lmdb, transforms3d, scipy.misc.imread/imsave and the
cffi-built `lib.utils.extend_utils._extend_utils`.

This is launcher / test infrastructure, not product code and not imported by it.  What the shims are:
* `easydict.EasyDict`  -- a real (small) implementation: `lib/utils/config.py` builds its whole configuration with it;
* `tensorboardX.SummaryWriter` -- a no-op recorder;
* `cv2.solvePnP` / `cv2.Rodrigues` -- backed by this repository's host PnP (pvnet_amd/pnp.py) so `lib.utils.evaluation_utils.pnp`
  works; every other attribute of a shimmed module is an inert placeholder that raises only when it is really USED for
  arithmetic (calling it returns another placeholder), which is enough for module-level imports and constants.

    import refshim; refshim.install(reference_root)      # then import / runpy the reference script
    python tools/run_reference.py tools/demo.py ...      # launcher: repo first on sys.path, shims, cwd = reference root
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

STUB_ROOTS = ("cv2", "torchvision", "plyfile", "skimage", "tensorboardX", "lmdb", "transforms3d", "OpenGL", "glumpy",
              "open3d", "cffi")


class _Anything:
    """inert placeholder: attribute access, calls, indexing and iteration all work and yield placeholders / nothing"""

    def __init__(self, name="stub"):
        object.__setattr__(self, "_name", name)

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(f"{self._name}.{k}")

    def __call__(self, *a, **k):
        return _Anything(f"{self._name}()")

    def __getitem__(self, k):
        return _Anything(f"{self._name}[]")

    def __iter__(self):
        return iter(())

    def __int__(self):
        return 0

    def __index__(self):
        return 0

    def __float__(self):
        return 0.0

    def __bool__(self):
        return False

    def __repr__(self):
        return f"<refshim placeholder {self._name}>"

    def __mro_entries__(self, bases):  # `class X(stub.Base):` works
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        v = _Anything(f"{self.__name__}.{k}")
        return v


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """serves `import root.any.sub.module` for the stubbed roots that are really missing"""

    def __init__(self, roots):
        self.roots = set(roots)

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class EasyDict(dict):
    """attribute-style dict (the subset of easydict the reference's config uses: recursive wrapping, get/set)"""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


class SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return lambda *a, **kw: None


def _missing(name):
    try:
        __import__(name)
        return False
    except Exception:
        return True


def install(reference_root=None):
    """register the shims for whatever is missing; idempotent.  Returns the list of shimmed roots."""
    sys.dont_write_bytecode = True  # importing from the (read-only) reference checkout must not drop __pycache__ there
    roots = [r for r in STUB_ROOTS if r not in sys.modules and _missing(r)]
    if roots:
        sys.meta_path.append(_StubFinder(roots))
    if "easydict" not in sys.modules and _missing("easydict"):
        m = types.ModuleType("easydict")
        m.EasyDict = EasyDict
        sys.modules["easydict"] = m
        roots.append("easydict")
    if "tensorboardX" in roots:
        import tensorboardX
        tensorboardX.SummaryWriter = SummaryWriter
    if "cv2" in roots:
        import cv2
        _wire_cv2(cv2)
    import scipy.misc as sm  # imread / imsave left SciPy in 1.2 (lib/utils/data_utils.py:20 imports them)
    for n in ("imread", "imsave", "imresize"):
        if not hasattr(sm, n):
            setattr(sm, n, _Anything(f"scipy.misc.{n}"))
    # (the cffi-built extension `lib.utils.extend_utils._extend_utils` -- Ceres + CUDA inside, extend_utils.py:3 -- is not
    # stubbed: this repository's overlay tree carries a real stand-in of that name on its native libraries)
    return roots


def _wire_cv2(cv2):
    """cv2.solvePnP / Rodrigues on this repository's host PnP, so that the reference's `evaluation_utils.pnp` runs"""
    import numpy as np

    def Rodrigues(x):
        from pvnet_amd import pnp as P
        x = np.asarray(x, np.float64)
        if x.size == 3:
            return P.rodrigues(x.reshape(3)), None
        return P.rodrigues_inv(x.reshape(3, 3)).reshape(3, 1), None

    def solvePnP(objectPoints, imagePoints, cameraMatrix, distCoeffs=None, rvec=None, tvec=None, useExtrinsicGuess=False,
                 flags=0, **kw):
        """cv2's argument list (the reference passes rvec / tvec / useExtrinsicGuess positionally, extend_utils.py:90-92).
        >= 6 points: this repository's pnp (linear start + LM).  4-5 points (the reference's SOLVEPNP_P3P start on the four
        best-weighted key-points): LM on the reprojection error from a fan of initial poses in front of the camera, the
        lowest cost wins -- only ever used as the start of the all-points refinement."""
        from pvnet_amd import pnp as P
        p3 = np.asarray(objectPoints, np.float64).reshape(-1, 3)
        p2 = np.asarray(imagePoints, np.float64).reshape(-1, 2)
        K = np.asarray(cameraMatrix, np.float64)
        if p3.shape[0] >= 6:
            pose = P.pnp(p3, p2, K)
            return True, P.rodrigues_inv(pose[:, :3]).reshape(3, 1), pose[:, 3].reshape(3, 1)
        size = np.linalg.norm(p3.max(0) - p3.min(0)) + 1e-12
        spread = np.linalg.norm(p2.max(0) - p2.min(0)) + 1e-12
        z0 = K[0, 0] * size / spread
        centre = np.linalg.solve(K, np.array([*p2.mean(0), 1.0])) * z0 - 0.0
        best = None
        for ax in np.eye(3):
            for ang in np.linspace(0, 2 * np.pi, 8, endpoint=False):
                for tilt in (0.0, 1.2, -1.2):
                    aa = ax * ang + np.roll(ax, 1) * tilt
                    R0 = P.rodrigues(aa)
                    x0 = np.concatenate([P.rodrigues_inv(R0), centre - R0 @ p3.mean(0)])
                    x = P._refine(x0, p3, p2, K, None, "native")
                    cost = float((P._residuals(x, p3, p2, K) ** 2).sum())
                    if np.isfinite(cost) and (best is None or cost < best[1]) and (P.rodrigues(x[:3]) @ p3.mean(0) + x[3:])[2] > 0:
                        best = (x, cost)
        x = best[0]
        return True, np.asarray(x[:3], np.float64).reshape(3, 1), np.asarray(x[3:], np.float64).reshape(3, 1)

    cv2.Rodrigues = Rodrigues
    cv2.solvePnP = solvePnP
    cv2.SOLVEPNP_ITERATIVE, cv2.SOLVEPNP_EPNP, cv2.SOLVEPNP_P3P = 0, 1, 2


REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pin_overlay(reference_root):
    """Make `lib.ransac_voting_gpu_layer.{ransac_voting_gpu, ransac_voting}` resolve to THIS repository's HIP layer for
    the rest of the process.  Path order alone is not enough for the reference's own scripts: `lib/utils/config.py:22-28`
    inserts the reference root at sys.path[0] while `tools/demo.py:5` is still importing, i.e. BEFORE `demo.py:8` imports
    the voting layer -- so the two overlay modules are imported here, first, and stay pinned in sys.modules (the `lib`
    namespace package itself keeps spanning both trees, so every other `lib.*` import still finds the reference)."""
    import importlib
    for p in (reference_root, REPO_ROOT):  # final order: repository first, reference second
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    mods = []
    for n in ("lib.ransac_voting_gpu_layer.ransac_voting", "lib.ransac_voting_gpu_layer.ransac_voting_gpu"):
        m = importlib.import_module(n)
        assert os.path.abspath(m.__file__).startswith(REPO_ROOT + os.sep), f"{n} resolved to {m.__file__}"
        mods.append(m)
    return mods


def import_reference_script(path, name=None, argv=None):
    """execute a reference script's module level UNCHANGED (its `if __name__ == "__main__"` block does not run) with the
    cwd it expects (its repository root: it opens `configs/linemod_train.json` relatively) and return the module."""
    import importlib.util
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference checkout
    path = os.path.abspath(path)
    root = os.path.dirname(os.path.dirname(path))
    name = name or "reference_" + os.path.splitext(os.path.basename(path))[0]
    old_cwd, old_argv = os.getcwd(), sys.argv
    os.chdir(root)
    sys.argv = [path] + list(argv or [])
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        os.chdir(old_cwd)
        sys.argv = old_argv
    return mod
