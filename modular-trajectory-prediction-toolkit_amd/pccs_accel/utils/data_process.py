"""Drop-in for the hot-path-adjacent functions of PCCSNet/utils/data_process.py
(transform_trajectory :111-122; the modality-loss label builders :134-220 live in
pccs_accel.scene)."""
from ..evaluation import transform_trajectory  # noqa: F401
