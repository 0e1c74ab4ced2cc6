"""Drop-in for PCCSNet/utils/checkings.py:4-22 (the two checks the clustering cache uses)."""
from ..checkings import check_cluster_consistency, check_memory_consistency  # noqa: F401
