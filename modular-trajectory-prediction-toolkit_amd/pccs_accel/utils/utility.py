"""Drop-in for the clustering functions of PCCSNet/utils/utility.py (imported by name at
PCCSNet/train_steps.py:10-12)."""
from ..clustering import cluster_data, get_memory_data, get_modality, k_means_cluster  # noqa: F401
