"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/mtp_accel.h declares, the ctypes table matches the header's arity, and the host-side
geometry helpers answer without a GPU.  No compute entry point is called here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mtp_accel.h")


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = re.findall(r"\b(?:const\s+char\s*\*\s*|int64_t\s+|int\s+)(mtp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)
    out = {}
    for name, args in protos:
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_header_declares_expected_entry_points():
    fns = _header_functions()
    for must in ("mtp_assign_f32", "mtp_centroid_sums_f32", "mtp_lloyd_f32", "mtp_cluster_means_f32",
                 "mtp_best_of_k_f32", "mtp_best_of_k_f64", "mtp_row_sqnorm_f32", "mtp_last_error"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from pccs_accel import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build() / make -C .../csrc"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_functions():
        assert hasattr(L, name), f"{name} declared in mtp_accel.h but not exported"


def test_ctypes_table_matches_header():
    from pccs_accel import _lib
    fns = _header_functions()
    assert set(_lib.SIGNATURES) == set(fns)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        assert len(argtypes) == fns[name], name


def test_geometry_helpers_without_gpu():
    from pccs_accel import _lib
    L = _lib.lib()
    assert L.mtp_version() >= 100
    # 200 centroids -> 8 tiles of 32 (padded to whole 2-tile LDS stages), 32*96+32 floats each
    assert L.mtp_staged_centers_floats(200, 96) == 8 * (32 * (96 + 4) + 32)
    assert L.mtp_staged_centers_floats(1024, 128) == 32 * (32 * (128 + 4) + 32)
    assert L.mtp_staged_centers_floats(5, 2) == 0  # generic kernel: no staged image
    assert L.mtp_lloyd_workspace_bytes(1024, 128) > 1024 * 128 * 8
    assert L.mtp_assign_workspace_bytes(200, 96) >= 8 * (32 * (96 + 4) + 32) * 4 + 200 * 4


def test_argument_errors_are_reported_not_faulted():
    """Bad arguments are rejected on the host before any launch, with a message."""
    from pccs_accel import _lib
    L = _lib.lib()
    rc = L.mtp_best_of_k_f32(None, None, 4, 5, 5, 12, 12.0, 0, None, None, None, None, None, None)
    assert rc == -1 and b"null" in L.mtp_last_error()
    rc = L.mtp_best_of_k_f32(None, None, 4, 5, 3, 12, 12.0, 0, None, None, None, None, None, None)
    assert rc == -1 and b"bad shape" in L.mtp_last_error()
    rc = L.mtp_assign_f32(None, 10, 96, None, 4, None, None, None, None, 0, None)
    assert rc == -1


def test_host_wrappers_refuse_cpu_tensors():
    import torch
    from pccs_accel import kernels as K
    from pccs_accel._lib import MtpError
    with pytest.raises(MtpError):
        K.assign(torch.zeros(4, 96), torch.zeros(2, 96))
    from pccs_accel.utils.metrics import displacement_error
    with pytest.raises(MtpError):
        displacement_error(torch.zeros(3, 12, 2), torch.zeros(1, 12, 2))
