"""pvnet_amd -- MI355X-native (gfx950) implementation of PVNet's RANSAC voting hot path.

Public surface (mirrors lib/ransac_voting_gpu_layer of zju3dv/pvnet):
    ransac_voting_layer_v3, generate_hypothesis, voting_for_hypothesis      (pvnet_amd.voting)
    sharded_ransac_voting_layer_v3                                          (pvnet_amd.distributed)
The compute lives in libpvnet_vote.so (HIP, C ABI in include/pvnet_vote.h); build it with
`python -m pvnet_amd.build`.
"""
from .voting import (generate_hypothesis, load_library, ransac_voting_layer_v3,  # noqa: F401
                     voting_for_hypothesis)

__all__ = ["ransac_voting_layer_v3", "generate_hypothesis", "voting_for_hypothesis", "load_library"]
