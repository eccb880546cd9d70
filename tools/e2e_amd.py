"""BASELINE.json configs[1] / [4] without the artefacts this image lacks (weights, LINEMOD, cv2, torchvision): what share
of a frame does the voting layer take next to the backbone?  SURVEY.md Appendix B: "cfg 2 degrades to GT vector field +
random-init backbone for timing" -- exactly that:

    image [b,3,480,640] -> stand-in ResNet-18-8s (plain PyTorch-ROCm / MIOpen, RANDOM weights, the reference's tensor
    interface: seg_pred [b,2,h,w], ver_pred [b,18,h,w], lib/networks/model_repository.py:76-78)
    -> EvalWrapper on the HIP layer (arg-max fused, tools/demo.py:46-55) -> host PnP per image (tools/demo.py:179)

A random backbone predicts a random mask, so its outputs are replaced (multiplied by zero, then added to) by the
synthetic ground-truth logits / field of the benchmark: the backbone is timed, the voting layer sees the workload of
BASELINE configs[2].  Timing only; no accuracy claim.      python tools/e2e_amd.py   (needs an MI355X)"""
import os
import sys
import time

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import pnp, synth, voting  # noqa: E402


def block(cin, cout, stride=1, dilation=1):
    class B(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = nn.Conv2d(cin, cout, 3, stride, dilation, dilation, bias=False)
            self.b1 = nn.BatchNorm2d(cout)
            self.c2 = nn.Conv2d(cout, cout, 3, 1, dilation, dilation, bias=False)
            self.b2 = nn.BatchNorm2d(cout)
            self.down = None
            if stride != 1 or cin != cout:
                self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

        def forward(self, x):
            y = torch.relu(self.b1(self.c1(x)))
            y = self.b2(self.c2(y))
            return torch.relu(y + (x if self.down is None else self.down(x)))
    return B()


class StandInResnet18_8s(nn.Module):
    """ResNet-18 trunk with output stride 8 (layers 3 and 4 dilated instead of strided) + the 8s -> 4s -> 2s -> raw
    decoder: the same layer shapes as the reference's Resnet18_8s, written from its description, random weights."""

    def __init__(self, ver_dim=18, seg_dim=2, fcdim=256, s8dim=128, s4dim=64, s2dim=32, raw_dim=32):
        super().__init__()
        self.seg_dim = seg_dim
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(True))
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.l1 = nn.Sequential(block(64, 64), block(64, 64))
        self.l2 = nn.Sequential(block(64, 128, 2), block(128, 128))
        self.l3 = nn.Sequential(block(128, 256, 1, 2), block(256, 256, 1, 2))
        self.l4 = nn.Sequential(block(256, 512, 1, 4), block(512, 512, 1, 4))
        self.fc = nn.Sequential(nn.Conv2d(512, fcdim, 3, 1, 1, bias=False), nn.BatchNorm2d(fcdim), nn.ReLU(True))

        def dec(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.LeakyReLU(0.1, True))
        self.conv8s, self.conv4s, self.conv2s = dec(128 + fcdim, s8dim), dec(64 + s8dim, s4dim), dec(64 + s4dim, s2dim)
        self.up = nn.UpsamplingBilinear2d(scale_factor=2)
        self.convraw = nn.Sequential(dec(3 + s2dim, raw_dim), nn.Conv2d(raw_dim, seg_dim + ver_dim, 1, 1))

    def forward(self, x):
        x2s = self.stem(x)
        x4s = self.l1(self.pool(x2s))
        x8s = self.l2(x4s)
        xfc = self.fc(self.l4(self.l3(x8s)))
        fm = self.up(self.conv8s(torch.cat([xfc, x8s], 1)))
        fm = self.up(self.conv4s(torch.cat([fm, x4s], 1)))
        fm = self.up(self.conv2s(torch.cat([fm, x2s], 1)))
        y = self.convraw(torch.cat([fm, x], 1))
        return y[:, :self.seg_dim], y[:, self.seg_dim:]


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    net = StandInResnet18_8s().to(dev).eval()
    head = voting.EvalWrapper(round_hyp_num=512, inlier_thresh=0.99)  # tools/demo.py:55
    X3 = np.random.default_rng(0).uniform(-0.08, 0.08, size=(9, 3))
    print("stand-in backbone parameters: %.1f M (random init); voting: 512 hypotheses, thresh 0.99, 9 key-points" %
          (sum(p.numel() for p in net.parameters()) / 1e6))
    for b, steps in ((1, 200), (8, 50), (32, 20)):
        mask, planar, _ = synth.make_batch(b, radius=40, noise=True, background="normal")
        m = torch.from_numpy(mask).to(dev).float()
        seg_gt = torch.stack([1.0 - m, m], 1).contiguous()
        ver_gt = torch.from_numpy(planar).to(dev)
        img = torch.randn(b, 3, 480, 640, device=dev)
        for amp, in_place in ((None, False), (torch.bfloat16, False), (torch.bfloat16, True)):
            def backbone():
                with torch.no_grad(), torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                    s, v = net(img)
                if in_place:  # round 2: the layer reads bf16 logits / fields where they lie (no .float() of 786 MB at b = 32)
                    return s * 0 + seg_gt.to(s.dtype), v * 0 + ver_gt.to(v.dtype)
                return s.float() * 0 + seg_gt, v.float() * 0 + ver_gt  # timing only: see the module docstring

            def frame(do_pnp):
                s, v = backbone()
                k = head(s, v)
                if do_pnp:
                    kc = k.cpu().numpy().astype(np.float64)  # the step's one host sync (tools/demo.py:176)
                    pnp.pnp_batch(X3, kc, pnp.LINEMOD_K)     # native DLT + LM for the b poses
                return k

            def timed(fn, n):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n

            def pipelined(n):
                """the pose solve of batch i overlaps the backbone of batch i + 1 -- key-points leave through a pinned,
                non-blocking copy + an event; the host solves batch i - 1 while the GPU works on batch i (tools/train_linemod.py:210-218
                solves inside the loop, behind a blocking .cpu())"""
                host = [torch.empty((b, 9, 2), dtype=torch.float32).pin_memory() for _ in range(2)]
                evs = [torch.cuda.Event() for _ in range(2)]
                poses = None

                def solve(j):
                    evs[j].synchronize()
                    return pnp.pnp_batch(X3, host[j].numpy().astype(np.float64), pnp.LINEMOD_K)
                for i in range(n):
                    s, v = backbone()
                    k = head(s, v)
                    host[i % 2].copy_(k, non_blocking=True)
                    evs[i % 2].record()
                    if i > 0:
                        poses = solve((i - 1) % 2)
                poses = solve((n - 1) % 2)
                return poses

            def timed_pipe(n):
                pipelined(3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pipelined(n)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n

            t_bb = timed(backbone, steps)
            s, v = backbone()
            t_vote = timed(lambda: head(s, v), steps)
            t_all = timed(lambda: frame(True), steps)
            t_pipe = timed_pipe(steps)
            label = "fp32" if amp is None else ("bf16, outputs read in place" if in_place else "bf16 autocast, .float()")
            print(f"b={b:2d} backbone {label:27s}: backbone {t_bb * 1e3:7.2f} ms  voting "
                  f"{t_vote * 1e3:6.3f} ms ({100 * t_vote / (t_bb + t_vote):4.1f} % of backbone+voting)  "
                  f"end to end with host PnP {t_all * 1e3:7.2f} ms = {b / t_all:8.1f} images/s; poses of batch i solved beside the "
                  f"backbone of batch i + 1: {t_pipe * 1e3:7.2f} ms = {b / t_pipe:8.1f} images/s", flush=True)


if __name__ == "__main__":
    main()
