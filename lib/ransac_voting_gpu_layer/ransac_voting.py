"""Stand-in for the reference's compiled extension module ``ransac_voting`` (src/ransac_voting.cpp:102-107):
same four function names and arities, backed by libpvnet_vote.so."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from pvnet_amd.voting import (generate_hypothesis, generate_hypothesis_vanishing_point,  # noqa: E402,F401
                              voting_for_hypothesis, voting_for_hypothesis_vanishing_point)
