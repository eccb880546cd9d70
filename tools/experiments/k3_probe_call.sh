python tools/experiments/k3_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
PVNET_SCORE_CULL=1 python -m pytest tests/test_disc_culling.py -m gpu -q 2>&1 | tail -3
python tools/cull_probe.py quick 2>&1 | grep -v amdgpu | cut -c1-330
