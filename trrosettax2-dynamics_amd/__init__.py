"""MI355X-native replacement of the trRosettaX2-Dynamics folding hot path (folding/folding.py + utils_ros).

The directory name carries a hyphen, so import it with importlib.import_module("trrosettax2-dynamics_amd")
or through the top-level alias module `trx2fold`.
"""
from . import protocol  # noqa: F401
from ._lib import Context, TERM_NAMES, load, make_params, make_runs  # noqa: F401
