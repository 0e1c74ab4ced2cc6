"""Drop-in for PCCSNet/utils/metrics.py:4-23."""
from ..metrics import displacement_error, final_displacement_error  # noqa: F401
