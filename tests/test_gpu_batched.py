"""The n_init restarts of KMeans.fit sharing every launch (mtp_kmeans_plusplus_batched_f32,
mtp_lloyd_batched_f32) must give, restart by restart, exactly what the one-restart entry points
give -- same seeds picked, same labels, centres, inertia and iteration counts -- including restarts
that pause for an empty-cluster relocation while the others carry on."""
import numpy as np
import pytest
import torch

import datagen
import kmeans_oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("n,d,k,b", [(3000, 96, 20, 5), (5000, 128, 33, 3), (2049, 64, 7, 20)])
def test_batched_seeding_equals_single(n, d, k, b, gpu_device):
    from pccs_accel import seeding
    X = datagen.make_encodings(200 + n, n, d, 12)
    Xc = dev(X - X.mean(axis=0), gpu_device)
    rs = np.random.RandomState(9)
    draws = [seeding.draw(rs, n, k) for _ in range(b)]
    cb, ib = seeding.kmeans_plusplus_batched(Xc, k, draws)
    rs = np.random.RandomState(9)
    for r in range(b):
        c1, i1 = seeding.kmeans_plusplus(Xc, k, rs)
        assert torch.equal(ib[r], i1) and torch.equal(cb[r], c1)


@pytest.mark.parametrize("n,d,k,b", [(3001, 96, 40, 1), (2500, 128, 25, 3), (4099, 64, 30, 6), (3333, 96, 64, 20),
                                     (1500, 96, 16, 23)])
def test_seeding_mfma_kernel_equals_fma_chain_kernel(n, d, k, b, gpu_device, monkeypatch):
    """The fp64-MFMA distance kernel (the default; both instantiations: 1-2 and 3-4 restarts per workgroup column, and more
    restarts than one column holds) against the multiply-add chain kernel it replaces (MTP_KPP_KERNEL=fma): same
    seeds, same centres.  (tools/mfma_f64_order checks the instruction's accumulation order itself.)"""
    from pccs_accel import seeding
    X = datagen.make_encodings(300 + n, n, d, 9)
    Xc = dev(X - X.mean(axis=0), gpu_device)
    rs = np.random.RandomState(4)
    draws = [seeding.draw(rs, n, k) for _ in range(b)]
    monkeypatch.setenv("MTP_KPP_KERNEL", "fma")
    c_fma, i_fma = seeding.kmeans_plusplus_batched(Xc, k, draws)
    torch.cuda.synchronize()
    monkeypatch.setenv("MTP_KPP_KERNEL", "mfma")
    c_mfma, i_mfma = seeding.kmeans_plusplus_batched(Xc, k, draws)
    torch.cuda.synchronize()
    assert torch.equal(i_fma, i_mfma) and torch.equal(c_fma, c_mfma)
    monkeypatch.delenv("MTP_KPP_KERNEL")  # the default: the fp64-MFMA kernel
    c_def, i_def = seeding.kmeans_plusplus_batched(Xc, k, draws)
    assert torch.equal(i_fma, i_def) and torch.equal(c_fma, c_def)
    # and against the CPU oracle's replay of sklearn, restart 0
    oc, oi = O.kmeans_plusplus(np.ascontiguousarray(Xc.cpu().numpy()), k, np.random.RandomState(4))
    assert np.array_equal(i_mfma[0].cpu().numpy(), oi)


@pytest.mark.parametrize("check_every", [1, 4])
def test_batched_lloyd_equals_single_with_relocation(check_every, gpu_device):
    from pccs_accel import kernels as K
    n, d, k, b = 4000, 96, 24, 6
    X = datagen.make_encodings(333, n, d, 10, 2000, 400)
    Xc = X - X.mean(axis=0)
    tol = float(O.tolerance(X))
    C0s = np.stack([Xc[datagen.make_init_indices(40 + r, n, k)] for r in range(b)]).copy()
    C0s[1, 5] = 50.0            # restart 1 and 4 start with unreachable centres -> relocation pauses
    C0s[4, 0] = -40.0
    C0s[4, 9] = 45.0
    lb, ib, cb, nb = K.lloyd_batched(dev(Xc, gpu_device), dev(C0s, gpu_device), tol_abs=tol, check_every=check_every)
    assert len(set(nb)) > 1      # the restarts really stop at different iterations
    for r in range(b):
        l1, i1, c1, n1 = K.lloyd(dev(Xc, gpu_device), dev(C0s[r], gpu_device), tol_abs=tol)
        assert nb[r] == n1 and torch.equal(lb[r], l1) and torch.equal(cb[r], c1) and float(ib[r]) == i1
        ol, oi, oc, on = O.kmeans_single_lloyd(Xc, C0s[r], 300, tol)
        assert on == n1 and O.classify_label_mismatches(Xc, oc, l1.cpu().numpy(), ol)["n_bad"] == 0


def test_batched_fit_equals_sequential_fit(gpu_device):
    from pccs_accel.clustering import kmeans_fit
    g = load_golden("g2_seeded_cluster_data")
    X = dev(datagen.make_encodings(int(g["seed"]), int(g["n"]), int(g["d"]), int(g["k_true"])), gpu_device)
    fits = [kmeans_fit(X, int(g["K"]), n_init=20, random_state=int(g["np_seed"]), batch=bt) for bt in (1, 7, 20)]
    for f in fits[1:]:
        assert torch.equal(f["labels"], fits[0]["labels"]) and torch.equal(f["centers"], fits[0]["centers"])
        assert f["inertia"] == fits[0]["inertia"] and f["n_iter"] == fits[0]["n_iter"]
    assert np.array_equal(fits[2]["labels"].cpu().numpy(), g["labels"])   # and the reference's seeded result


def test_batched_max_iter(gpu_device):
    from pccs_accel import kernels as K
    g = load_golden("g1_hard_n5000_d96_k64")
    from conftest import g1_inputs
    X, C0 = g1_inputs(g)
    Xc, C0c = X - g["x_mean"], C0 - g["x_mean"]
    C0s = np.stack([C0c, C0c[::-1].copy()])
    lb, ib, cb, nb = K.lloyd_batched(dev(Xc, gpu_device), dev(C0s, gpu_device), max_iter=3, tol_abs=float(g["tol_abs"]))
    assert nb == [3, 3]
    for r in range(2):
        assert np.array_equal(lb[r].cpu().numpy(), O.assign(Xc, cb[r].cpu().numpy()))  # labels match final centres


@pytest.mark.parametrize("n,d,k,b", [(4000, 96, 24, 6), (3001, 128, 37, 5), (2500, 64, 8, 20), (6007, 96, 200, 7),
                                     (2049, 96, 33, 23), (1500, 96, 400, 3)])
def test_restart_groups_do_not_change_the_result(n, d, k, b, gpu_device, monkeypatch):
    """mtp_lloyd_batched_f32 with the restarts' centres grouped into shared images (k_assign_mfma_grouped: a workgroup
    scores its points against R restarts, restart boundaries between accumulator register quads) against one image
    per restart (k_assign_mfma): every group size -- boundaries at every row offset that a multiple of kp takes in a
    32-row tile, k not a multiple of 8 (padding rows inside a group), a last group with fewer restarts, more restarts
    than one group may hold, restarts that pause for a relocation or stop early while their group carries on -- gives
    the same labels, centres, inertia and iteration counts, bit for bit; and restart 0 those of the single run."""
    from pccs_accel import kernels as K
    X = datagen.make_encodings(900 + n, n, d, 10, 2000, 500)
    Xc = X - X.mean(axis=0)
    tol = float(O.tolerance(X))
    C0s = np.stack([Xc[datagen.make_init_indices(60 + r, n, k)] for r in range(b)]).copy()
    C0s[1, 3] = 50.0             # an unreachable centre: restart 1 pauses for a relocation while its group runs on
    if b > 4:
        C0s[4, 0] = -40.0
    Xd, Cd = dev(Xc, gpu_device), dev(C0s, gpu_device)
    monkeypatch.setenv("MTP_RESTART_GROUP", "0")
    ref = K.lloyd_batched(Xd, Cd, tol_abs=tol, check_every=3)
    l1, i1, c1, n1 = K.lloyd(Xd, dev(C0s[0], gpu_device), tol_abs=tol)
    assert ref[3][0] == n1 and torch.equal(ref[0][0], l1) and torch.equal(ref[2][0], c1)
    assert len(set(ref[3])) > 1  # the restarts really stop at different iterations
    for group in (2, 3, 4, 5, 8, 32):
        monkeypatch.setenv("MTP_RESTART_GROUP", str(group))
        for check_every in (1, 4):
            got = K.lloyd_batched(Xd, Cd, tol_abs=tol, check_every=check_every)
            assert got[3] == ref[3], (group, got[3], ref[3])
            assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]) and torch.equal(got[1], ref[1]), group
    monkeypatch.delenv("MTP_RESTART_GROUP")  # the library's own choice
    got = K.lloyd_batched(Xd, Cd, tol_abs=tol)
    assert got[3] == ref[3] and torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
