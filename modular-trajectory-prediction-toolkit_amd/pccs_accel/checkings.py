"""Cache-staleness checks for the clustering artefacts.

Same contract as PCCSNet/utils/checkings.py:4-22: ``memory.npy`` must be newer than both
encoder checkpoints, ``cluster_result.npy`` newer than ``memory.npy`` and sized for
``args.n_cluster``; ``args.no_time_check`` bypasses both.
"""
import os


def _ctime(save_dir, name):
    return os.path.getctime(os.path.join(save_dir, name))


def check_memory_consistency(save_dir, args):
    if args.no_time_check:
        return True
    memory_time = _ctime(save_dir, "memory.npy")
    return memory_time > _ctime(save_dir, "Obs_Encoder.pth") and memory_time > _ctime(save_dir, "Pred_Encoder.pth")


def check_cluster_consistency(save_dir, size_array, args):
    if args.no_time_check:
        return True
    if len(size_array) != args.n_cluster:
        return False
    return _ctime(save_dir, "cluster_result.npy") >= _ctime(save_dir, "memory.npy")


def check_prob_consistency(save_dir, args):
    """PCCSNet/utils/checkings.py:25-32: probabilities.pkl must not be older than cluster_result.npy."""
    if args.no_time_check:
        return True
    return not (_ctime(save_dir, "cluster_result.npy") > _ctime(save_dir, "probabilities.pkl"))
