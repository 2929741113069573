"""Re-export: the checkpoint layout table and the synthetic-checkpoint recipe live in
``pointstowood_amd/synthetic_weights.py`` (data generators, not oracle arithmetic); kept for the tests and the golden
generator (``tests/golden/make_golden.py``)."""
from pointstowood_amd.synthetic_weights import *  # noqa: F401,F403
from pointstowood_amd.synthetic_weights import BN_EPS  # noqa: F401
