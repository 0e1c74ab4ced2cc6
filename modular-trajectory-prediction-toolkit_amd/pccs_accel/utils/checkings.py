"""Drop-in for PCCSNet/utils/checkings.py:4-32 (the checks the clustering / label caches use)."""
from ..checkings import check_cluster_consistency, check_memory_consistency, check_prob_consistency  # noqa: F401
