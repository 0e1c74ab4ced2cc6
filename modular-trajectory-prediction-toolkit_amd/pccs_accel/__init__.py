"""pccs_accel -- MI355X-native drop-in for PCCSNet's modality-clustering / best-of-K path.

    from pccs_accel.utils.utility import get_memory_data, cluster_data, k_means_cluster, get_modality
    from pccs_accel.utils.metrics import displacement_error, final_displacement_error
    from pccs_accel.evaluation import evaluate, best_of_k

The arithmetic runs in libmtp_accel.so (hand-written gfx950 HIP kernels, include/mtp_accel.h);
importing this package without that library, or calling it without a GPU, raises.
"""
from ._lib import LIB_PATH, MtpError, lib  # noqa: F401

__all__ = ["LIB_PATH", "MtpError", "lib"]
