"""Drop-in for the hot-path-adjacent functions of PCCSNet/utils/data_process.py:
transform_trajectory (:111-122) and the modality-loss label builders (:134-220)."""
from ..evaluation import transform_trajectory  # noqa: F401
from ..scene import gen_prob, get_scene_data, process_scene  # noqa: F401
