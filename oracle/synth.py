"""Re-export: the synthetic voxel generators live in ``pointstowood_amd/synthetic_voxels.py`` (they are input
generators, not oracle arithmetic); kept so that ``from oracle import synth`` in the tests keeps working."""
from pointstowood_amd.synthetic_voxels import *  # noqa: F401,F403
from pointstowood_amd.synthetic_voxels import _finish  # noqa: F401
