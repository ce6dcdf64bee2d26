"""Import shim: lets the reference driver's ``from BundleAdjustment import *``
(``Reconstruction.py:9``) resolve to the MI355X engine when this directory is on ``sys.path``."""
from structure_from_motion_amd.BundleAdjustment import *  # noqa: F401,F403
from structure_from_motion_amd.BundleAdjustment import __all__  # noqa: F401
