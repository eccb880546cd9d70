"""Imports the reference's OWN tools/demo.py and tools/train_linemod.py UNCHANGED (through tools/refshim.py) and records
how their EvalWrapper classes call the voting layer -- row N1 of VERDICT r01 ("reference callers run unchanged").

    python tools/reference_callers_probe.py /path/to/pvnet [--device cuda]     -> one JSON document on stdout

With --device cpu (default; the build container has no GPU) the five voting-layer entry points the scripts bound at
import are first checked to BE this repository's HIP functions, then swapped for recorders, and every wrapper's
forward() is run on small CPU tensors: the JSON holds, per wrapper, the function it called, the non-tensor arguments,
and dtype / shape / strides of the tensors it passed (tests/golden/reference_callers.json is this output; the GPU test
replays it against the HIP layer).  With --device cuda the wrappers run for real on the demo fixture's ground-truth
field and the JSON holds their outputs.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import refshim  # noqa: E402


def tensor_spec(t):
    return {"dtype": str(t.dtype), "shape": list(t.shape), "stride": list(t.stride()), "contiguous": t.is_contiguous()}


def main(argv):
    ref = os.path.abspath(argv[0])
    device = argv[argv.index("--device") + 1] if "--device" in argv else "cpu"
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # the reference's modules print while importing
    refshim.install(ref)
    refshim.pin_overlay(ref)
    import numpy as np
    import torch
    from pvnet_amd import voting
    demo = refshim.import_reference_script(os.path.join(ref, "tools", "demo.py"))
    train = refshim.import_reference_script(os.path.join(ref, "tools", "train_linemod.py"),
                                            argv=["--cfg_file", "configs/linemod_train.json", "--linemod_cls", "cat"])
    bound = {
        "demo.ransac_voting_layer_v3": demo.ransac_voting_layer_v3 is voting.ransac_voting_layer_v3,
        "train.ransac_voting_layer_v3": train.ransac_voting_layer_v3 is voting.ransac_voting_layer_v3,
        "train.ransac_voting_layer_v5": train.ransac_voting_layer_v5 is voting.ransac_voting_layer_v5,
        "train.estimate_voting_distribution_with_mean":
            train.estimate_voting_distribution_with_mean is voting.estimate_voting_distribution_with_mean,
        "train.ransac_motion_voting": train.ransac_motion_voting is voting.ransac_motion_voting,
    }
    out = {"reference_root": ref, "bound_to_hip_layer": bound, "device": device, "calls": {}}
    wrappers = [("demo.EvalWrapper", demo, demo.EvalWrapper, {}),
                ("train.EvalWrapper", train, train.EvalWrapper, {}),
                ("train.EvalWrapper[use_uncertainty]", train, train.EvalWrapper, {"use_uncertainty": True}),
                ("train.MotionEvalWrapper", train, train.MotionEvalWrapper, {}),
                ("train.UncertaintyEvalWrapper", train, train.UncertaintyEvalWrapper, {})]
    names = ("ransac_voting_layer_v3", "ransac_voting_layer_v5", "estimate_voting_distribution_with_mean",
             "ransac_motion_voting")
    if device == "cpu":
        b, vn, h, w = 2, 9, 24, 32
        g = torch.Generator().manual_seed(0)
        seg_pred = torch.randn((b, 2, h, w), generator=g)
        vertex_pred = torch.randn((b, 2 * vn, h, w), generator=g)
        for label, mod, cls, kw in wrappers:
            log = []

            def recorder(fname):
                def f(*a, **k):
                    log.append({"function": fname,
                                "tensor_args": [tensor_spec(x) for x in a if isinstance(x, torch.Tensor)],
                                "args": [x for x in a if not isinstance(x, torch.Tensor)],
                                "kwargs": {kk: vv for kk, vv in k.items() if not isinstance(vv, torch.Tensor)},
                                "tensor_kwargs": {kk: tensor_spec(vv) for kk, vv in k.items() if isinstance(vv, torch.Tensor)}})
                    pts = torch.zeros((b, vn, 2))
                    if fname == "ransac_voting_layer_v5":
                        return pts, torch.zeros((b, vn))
                    if fname == "estimate_voting_distribution_with_mean":
                        return a[2], torch.zeros((b, vn, 2, 2))
                    return pts
                return f

            saved = {n: getattr(mod, n) for n in names if hasattr(mod, n)}
            for n in saved:
                setattr(mod, n, recorder(n))
            try:
                cls()(seg_pred, vertex_pred, **kw)
            finally:
                for n, f in saved.items():
                    setattr(mod, n, f)
            out["calls"][label] = {"forward_kwargs": kw, "calls": log}
        out["input"] = {"seg_pred": tensor_spec(seg_pred), "vertex_pred": tensor_spec(vertex_pred)}
    else:
        sys.path.insert(0, ROOT)
        from pvnet_amd import synth
        g = np.load(os.path.join(ROOT, "tests", "golden", "demo_cat.npz"))
        hh, ww = (int(x) for x in g["shape"])
        mask = np.unpackbits(g["mask_bits"])[: hh * ww].reshape(hh, ww)
        planar = synth.field_from_keypoints(mask.astype(bool), g["points_2d"])
        dev = torch.device("cuda:0")
        m = torch.from_numpy(mask.astype(np.int64)).to(dev)
        seg_pred = torch.stack([1.0 - m.float(), m.float()])[None].contiguous()  # logits whose arg-max is the mask
        vertex_pred = torch.from_numpy(planar[None]).to(dev)
        for label, mod, cls, kw in wrappers:
            torch.manual_seed(7)
            r = cls().to(dev)(seg_pred, vertex_pred, **kw)
            r = r if isinstance(r, tuple) else (r,)
            out["calls"][label] = {"forward_kwargs": kw, "outputs": [x.detach().cpu().numpy().tolist() for x in r]}
        out["points_2d"] = g["points_2d"].tolist()
    os.write(real_stdout, (json.dumps(out, indent=1, sort_keys=True) + "\n").encode())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
