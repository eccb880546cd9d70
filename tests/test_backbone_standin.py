"""The backbone half of the end-to-end timing (tools/e2e_amd.py) uses a stand-in written from the layer shapes, because the
reference's Python cannot travel to the GPU box.  HERE, where /root/reference exists, the stand-in is pinned against the reference's
own class: `Resnet18_8s` (lib/networks/model_repository.py:7-80, with `model_zoo.load_url` of lib/networks/resnet.py:231 stubbed --
there is no network) must have the same parameters and buffers in the same order, and with the stand-in's weights copied in, the
same outputs.  So what tools/e2e_amd.py times IS the reference's network (random weights), layer for layer."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r"""
import sys, types, torch
sys.path.insert(0, %(tools)r); sys.path.insert(0, %(root)r)
import refshim
refshim.install(%(ref)r)
import e2e_amd                                   # the stand-in (imports the repo's own packages first)
for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
    del sys.modules[k]                           # the repo's drop-in overlay package is also called `lib`
sys.path.insert(0, %(ref)r)
import lib.networks.resnet as R
R.model_zoo.load_url = lambda *a, **k: {}        # resnet.py:231 -- no network here
R.ResNet.load_state_dict = lambda self, sd, *a, **k: None
from lib.networks.model_repository import Resnet18_8s
torch.manual_seed(0)
ref = Resnet18_8s(ver_dim=18, seg_dim=2).eval()
mine = e2e_amd.StandInResnet18_8s(ver_dim=18, seg_dim=2).eval()
# the reference keeps ResNet's ImageNet head behind `fc` replaced (model_repository.py:23-27) and nothing else unused
rp = [(n, p) for n, p in ref.named_parameters()]
mp = [(n, p) for n, p in mine.named_parameters()]
rb = [(n, b) for n, b in ref.named_buffers()]
mb = [(n, b) for n, b in mine.named_buffers()]
assert [tuple(p.shape) for _, p in rp] == [tuple(p.shape) for _, p in mp], "parameter shapes / order differ"
assert [tuple(b.shape) for _, b in rb] == [tuple(b.shape) for _, b in mb], "buffer shapes / order differ"
with torch.no_grad():
    for m in mine.modules():   # non-trivial statistics and affine terms, so that every BatchNorm matters
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    for (_, a), (_, b) in zip(rp, mp):
        a.copy_(b)
    for (_, a), (_, b) in zip(rb, mb):
        a.copy_(b)
    x = torch.randn(2, 3, 64, 96)
    s0, v0 = ref(x)
    s1, v1 = mine(x)
print("params", sum(p.numel() for _, p in rp), "max |d seg|", float((s0 - s1).abs().max()), "max |d ver|", float((v0 - v1).abs().max()))
assert s0.shape == s1.shape == (2, 2, 64, 96) and v0.shape == v1.shape == (2, 18, 64, 96)
assert float(s0.abs().max()) > 1e-3 and float(v0.std()) > 1e-3
assert torch.allclose(s0, s1, atol=1e-5, rtol=1e-5) and torch.allclose(v0, v1, atol=1e-5, rtol=1e-5)
print("STANDIN_EQUALS_REFERENCE")
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present (GPU box)")
def test_standin_backbone_is_the_references_resnet18_8s():
    code = SCRIPT % dict(tools=os.path.join(ROOT, "tools"), root=ROOT, ref=REF)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "STANDIN_EQUALS_REFERENCE" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
