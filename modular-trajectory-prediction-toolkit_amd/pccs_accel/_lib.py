"""ctypes binding of libmtp_accel.so (include/mtp_accel.h).

There is no fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmtp_accel.so")

STATE_RUNNING, STATE_STRICT, STATE_TOL, STATE_RELOCATE, STATE_MAXITER = 0, 1, 2, 3, 4
FLAG_STATE, FLAG_N_ITER, FLAG_N_EMPTY, FLAG_WORDS = 0, 1, 2, 8
POT_F64, POT_OPENBLAS = 0, 1  # mtp_accel.h MTP_POT_*
LLOYD_NO_FINAL_ESTEP = 1
LLOYD_SUMS_SKLEARN_1T = 2

_vp, _i64, _i32, _f32, _f64, _sz = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_double,
                                    ctypes.c_size_t)

# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "mtp_last_error": (ctypes.c_char_p, []),
    "mtp_version": (_i32, []),
    "mtp_staged_centers_floats": (_i64, [_i32, _i32]),
    "mtp_assign_workspace_bytes": (_i64, [_i32, _i32]),
    "mtp_lloyd_workspace_bytes": (_i64, [_i32, _i32]),
    "mtp_row_sqnorm_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "mtp_col_sums_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "mtp_col_sqdev_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "mtp_col_seqsum_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "mtp_col_seqsum_carry_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "mtp_sub_rowvec_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "mtp_assign_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mtp_assign_noalloc_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp]),
    "mtp_centroid_sums_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp]),
    "mtp_cluster_means_f32": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "mtp_label_targets_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "mtp_inertia_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mtp_lloyd_begin_f32": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mtp_lloyd_estep_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "mtp_lloyd_assign_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "mtp_lloyd_accumulate_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp]),
    "mtp_lloyd_mstep_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _f64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mtp_far_points_scratch_bytes": (_i64, [_i64]),
    "mtp_far_points_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mtp_apply_relocation_f32": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mtp_pairwise_sumsq_f32": (_i32, [_vp, _i32, _vp, _vp]),
    "mtp_far_merge_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mtp_lloyd_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _f64, _i32, _vp, _vp, ctypes.POINTER(ctypes.c_int), _i32,
                             _vp, _sz, _vp]),
    "mtp_comm_unique_id": (_i32, [_vp]),
    "mtp_comm_create": (_i32, [_i32, _i32, _vp, ctypes.POINTER(_vp)]),
    "mtp_comm_destroy": (_i32, [_vp]),
    "mtp_lloyd_sharded_workspace_bytes": (_i64, [_i32, _i32, _i64, _i32]),
    "mtp_lloyd_sharded_f32": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _i32, _i32, _f64, _i32, _vp, _vp,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float), _i32, _vp, _sz, _vp]),
    "mtp_lloyd_batched_workspace_bytes": (_i64, [_i32, _i32, _i32, _i64]),
    "mtp_lloyd_batched_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _i32, _f64, _i32, _vp, _vp,
                                     ctypes.POINTER(ctypes.c_int), _i32, _vp, _sz, _vp]),
    "mtp_kpp_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "mtp_kpp_batched_workspace_bytes": (_i64, [_i64, _i32, _i32, _i32]),
    "mtp_kmeans_plusplus_batched_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mtp_kmeans_plusplus_f32": (_i32, [_vp, _i64, _i32, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mtp_blas_order_rowsums_workspace_bytes": (_i64, [_i64, _i32]),
    "mtp_blas_order_rowsums_f32": (_i32, [_vp, _i32, _i64, _vp, _vp, _sz, _vp]),
    "mtp_blas_order_blocks": (_i64, [_i64]),
    "mtp_blas_order_block_sums_f32": (_i32, [_vp, _i32, _i64, _i32, _vp, _f64, _vp, _vp]),
    "mtp_blas_order_combine_f32": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "mtp_blas_order_sdot_carry_f32": (_i32, [_vp, _i64, _i32, _i64, _i64, _vp, _i32, _vp, _vp]),
    "mtp_kpp_candidate_dists_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "mtp_kpp_count_below_workspace_bytes": (_i64, [_i64, _i32]),
    "mtp_kpp_count_below_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "mtp_kpp_row_sqnorm_f64": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "mtp_kpp_candidate_dists_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mtp_best_of_k_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mtp_best_of_k_f64": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mtp_ade_fde_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "mtp_speed_angle_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "mtp_scene_neighbors_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp]),
    "mtp_soft_labels_f64": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "mtp_decode_trajectories_f32": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mtp_best_of_k_disp_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mtp_nms_rerank_f32": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _f64, _f64, _vp, _vp, _vp]),
    "mtp_nms_rerank_f64": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _f64, _f64, _vp, _vp, _vp]),
    "mtp_gather_candidates_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "mtp_gather_candidates_f64": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "mtp_best_of_k_reranked_f32": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp,
                                          _vp, _vp]),
    "mtp_best_of_k_reranked_f64": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _f64, _i32, _vp, _vp, _vp, _vp,
                                          _vp, _vp]),
}

_LIB = None


class MtpError(RuntimeError):
    pass


def lib():
    """Load libmtp_accel.so (once).  Raises if it has not been built -- there is no CPU path."""
    global _LIB
    if _LIB is None:
        # torch first: its bundled HIP runtime must be the one this process initialises.  Loading
        # libmtp_accel.so before torch pulls in /opt/rocm's libamdhip64 as a second runtime, and that
        # one then reports "no ROCm-capable device" when torch already owns the GPU.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise MtpError(
                f"{LIB_PATH} not found: build it with `make -C modular-trajectory-prediction-toolkit_amd/csrc` "
                "(or __graft_entry__.build()); pccs_accel has no fallback path")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        raise MtpError(f"{what} failed (rc={rc}): {lib().mtp_last_error().decode(errors='replace')}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
