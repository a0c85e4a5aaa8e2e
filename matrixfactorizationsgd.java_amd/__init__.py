"""MI355X-native matrix-factorisation SGD trainer.

Host-side mirror of the (absent) reference's ``MatrixFactorizationSGD``
``train()/predict()`` surface over ``libmfsgd.so`` (hand-written HIP for gfx950
behind the C-ABI of ``include/mfsgd.h``).  There is no CPU compute path in this
package: without the built library, or without a gfx950 device, compute calls
raise.
"""
from .trainer import MatrixFactorizationSGD, MfsgdError, dsgd_plan, dsgd_plan_ex  # noqa: F401
from ._lib import load_library, library_path  # noqa: F401
from . import synth  # noqa: F401
from .datasets import load_ratings  # noqa: F401

__all__ = ["MatrixFactorizationSGD", "MfsgdError", "dsgd_plan", "dsgd_plan_ex", "load_library", "library_path", "synth", "load_ratings"]
