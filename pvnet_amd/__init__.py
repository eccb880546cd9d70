"""pvnet_amd -- MI355X-native (gfx950) implementation of PVNet's RANSAC voting hot path.

Public surface (mirrors lib/ransac_voting_gpu_layer of zju3dv/pvnet):
    ransac_voting_layer_v3, generate_hypothesis, voting_for_hypothesis      (pvnet_amd.voting)
    sharded_ransac_voting_layer_v3                                          (pvnet_amd.distributed)
The compute lives in libpvnet_vote.so (HIP, C ABI in include/pvnet_vote.h); build it with
`python -m pvnet_amd.build`.
"""
from .voting import (estimate_voting_distribution_with_mean, generate_hypothesis,  # noqa: F401
                     generate_hypothesis_counts, load_library, ransac_motion_voting, ransac_voting_layer_v3,
                     ransac_voting_layer_v5, voting_for_hypothesis)

__all__ = ["ransac_voting_layer_v3", "generate_hypothesis", "voting_for_hypothesis", "load_library"]
