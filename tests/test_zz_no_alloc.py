"""The allocation-free entry point ``mtp_assign_noalloc_f32`` (include/mtp_accel.h) against the entry
point it is a variant of, ``mtp_assign_f32``: same labels and the same "labels changed" count, bit for
bit, on two calls in a row that share the library's built-in scratch buffer -- the second with a smaller
input (fewer points, fewer centroids, another feature size), so anything the first call left in the
buffer would show.  Both go through the C-ABI (ctypes); sizes beyond the buffer are refused."""
import numpy as np
import pytest
import torch

import datagen
from pccs_accel import kernels as K
from pccs_accel._lib import check, lib, ptr, stream_ptr

pytestmark = pytest.mark.gpu

NOALLOC_BYTES = 1 << 20  # MTP_NOALLOC_SCRATCH_BYTES


def _assign_noalloc(X, C, labels, n_changed):
    n, d = X.shape
    with torch.cuda.device(X.device):
        rc = lib().mtp_assign_noalloc_f32(ptr(X), n, d, ptr(C), C.shape[0], ptr(labels), ptr(n_changed), stream_ptr())
    return rc


def _case(dev, seed, n, d, k):
    X = datagen.make_encodings(seed, n, d, max(k // 2, 2), 1800, 600)
    C = X[datagen.make_init_indices(seed + 1, n, k)].copy()
    return torch.from_numpy(X).to(dev), torch.from_numpy(C).to(dev)


def test_noalloc_matches_assign_on_two_calls_in_a_row(gpu_device):
    dev = gpu_device
    # first call: the largest shape the buffer is documented for; second: smaller in every dimension;
    # third: the generic (non-MFMA) kernel, which keeps only ||c||^2 in the buffer
    for seed, n, d, k in ((901, 20000, 128, 1024), (903, 5003, 96, 200), (905, 777, 10, 37)):
        X, C = _case(dev, seed, n, d, k)
        prev = torch.from_numpy(np.random.RandomState(seed).randint(0, k, size=n).astype(np.int32)).to(dev)
        want = prev.clone()
        want_changed = torch.zeros(1, dtype=torch.float64, device=dev)
        K.assign(X, C, labels=want, n_changed=want_changed)

        got = prev.clone()
        got_changed = torch.zeros(1, dtype=torch.float64, device=dev)
        check(_assign_noalloc(X, C, got, got_changed), "mtp_assign_noalloc_f32")
        torch.cuda.synchronize()
        assert torch.equal(got, want), (n, d, k, int((got != want).sum()))
        assert float(got_changed.item()) == float(want_changed.item())
        assert int(got.min()) >= 0 and int(got.max()) < k


def test_noalloc_refuses_sizes_beyond_its_buffer(gpu_device):
    dev = gpu_device
    k, d = 4096, 128
    assert int(lib().mtp_assign_workspace_bytes(k, d)) > NOALLOC_BYTES
    X = torch.zeros((64, d), dtype=torch.float32, device=dev)
    C = torch.zeros((k, d), dtype=torch.float32, device=dev)
    labels = torch.zeros(64, dtype=torch.int32, device=dev)
    rc = _assign_noalloc(X, C, labels, None)
    assert rc == -3 and b"built-in buffer" in lib().mtp_last_error()
