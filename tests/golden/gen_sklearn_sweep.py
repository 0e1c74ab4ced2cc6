"""Generate tests/golden/g30_sklearn_sweep.npz: scikit-learn's own KMeans on ONE thread over seeded random shapes.

Run ONLY in the build container (needs scikit-learn 1.7.2, the third-party package that holds the reference's
k-means arithmetic, PCCSNet/utils/utility.py:121):

    python tests/golden/gen_sklearn_sweep.py [count] [seed]            # G30, 240 cases from seed 30
    python tests/golden/gen_sklearn_sweep.py [count] [seed] init       # G31, 160 cases from seed 31
    python tests/golden/gen_sklearn_sweep.py 0 0 singleton             # G36, one constructed case
    python tests/golden/gen_sklearn_sweep.py [count] [seed] kpp        # G37, 120 cases from seed 37
    python tests/golden/gen_sklearn_sweep.py [count] [seed] multi      # G38, 120 cases from seed 38

For every case it draws (n, d, K, n_init, random_state) and the points (datagen's integer mixture, every other time
scaled and jittered into floats with full mantissas), runs ``KMeans(K, n_init=n_init, random_state=s).fit(X)`` under
``threadpool_limits(1)`` -- the deterministic run that the product's ``sums="sklearn_1t"`` mode reproduces bit for bit
(fixture G15 holds six reference-sized ones) -- and stores the parameters, a digest of the points, and of the result:
n_iter, the inertia's bits, digests of the labels and of the centres, the cluster sizes' digest.  Data only.

``init`` (fixture G31): ``KMeans(K, init=C0, n_init=1).fit(X)`` from given centres instead -- rows of X, one of them
moved out of reach, so that its cluster starts empty and sklearn relocates it to the point farthest from its centre
(_k_means_common.pyx:165-200; ONE empty cluster at a time: with more, np.argpartition leaves their order to the
CPU's SIMD selection kernel) -- over points a third of which may be exact duplicates of other rows (equal distances
in the E-step: the lowest label wins, _k_means_lloyd.pyx:197-199).

``singleton`` (fixture G36): a relocation that takes a cluster's ONLY member.  Cluster 1 starts empty, the farthest
point is the single member of cluster 0, so cluster 0 is left without weight and ``_average_centers``
(_k_means_common.pyx:280-296, in place in index order) gives it the row of the heaviest cluster as it stands -- the
heaviest is cluster 2, not yet averaged when j = 0 is served: its raw sum.  ``KMeans(3, init=C0, n_init=1,
max_iter=it)`` for it = 1, 2, 3 on one thread; inputs and results stored whole (51 x 4).  The mirrored case (the
heaviest cluster first) is stored beside it.

``kpp`` (fixture G37): sklearn's public ``kmeans_plusplus(X, K, random_state=s)`` alone (cluster/_kmeans.py:174-272)
on one thread: K from 1 to 60 and, every fifth case, 410 / 700 / 1100 / 1250 (8 and 9 local trials), feature sizes on
and off the MFMA path, centred full-mantissa points.  The K chosen row indices, whole.

``multi`` (fixture G38): G31 with SEVERAL centres out of reach, so that sklearn relocates several empty clusters in one
iteration.  Which far point goes to which empty cluster is then left to ``np.argpartition``'s SIMD selection kernel
(unspecified; sklearn's own test accepts either outcome), so the result is stored up to the names of the clusters:
n_iter, the inertia's bits (a chain over the points: independent of the names), and digests of the labels renamed
in order of first appearance, of the sorted sizes and of the centres sorted as rows.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import datagen  # noqa: E402


def crc(a, dtype):
    return zlib.crc32(np.ascontiguousarray(a, dtype=dtype).tobytes())


def draw_case(seed):
    """The parameters and points of case ``seed``; the same in the generator and in the test."""
    rs = np.random.RandomState(seed)
    n = int(rs.randint(60, 9000))
    d = int(rs.choice([64, 96, 128, 96, int(rs.randint(2, 41)), 4 * int(rs.randint(1, 60))]))
    k = int(min(rs.randint(2, 120), n // 3))
    n_init = int(rs.choice([1, 2, 3, 5]))
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, k // 2), 1500, 700)
    if rs.rand() < 0.5:
        X = X * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(X.shape) * 1e-3).astype(np.float32)
    X = np.ascontiguousarray(X.astype(np.float32))
    return X, k, n_init, int(rs.randint(1 << 30))


def draw_init_case(seed):
    """(X, C0): the points and initial centres of G31's case ``seed``."""
    rs = np.random.RandomState(seed)
    n = int(rs.randint(60, 7000))
    d = int(rs.choice([64, 96, 128, 96, int(rs.randint(2, 41)), 4 * int(rs.randint(1, 60))]))
    k = int(min(rs.randint(3, 100), n // 3))
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, k // 2), 1500, 700)
    if rs.rand() < 0.5:
        X = X * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(X.shape) * 1e-3).astype(np.float32)
    X = np.ascontiguousarray(X.astype(np.float32))
    if rs.rand() < 0.4:  # exact duplicates of other rows
        m = n // 3
        X[rs.randint(0, n, size=m)] = X[rs.randint(0, n, size=m)]
    C0 = X[rs.choice(n, size=k, replace=False)].copy()
    if rs.rand() < 0.7:
        C0[int(rs.randint(k))] += np.float32(50.0)
    return X, C0


def main_init(count, seed0):
    import sklearn
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    rec = {key: [] for key in ("n", "d", "k", "x_crc", "n_iter", "inertia", "labels_crc", "centers_crc", "sizes_crc")}
    with threadpool_limits(limits=1):
        for i, s in enumerate(seeds):
            X, C0 = draw_init_case(s)
            k = C0.shape[0]
            km = KMeans(n_clusters=k, init=C0.copy(), n_init=1).fit(X)
            again = KMeans(n_clusters=k, init=C0.copy(), n_init=1).fit(X)
            assert np.array_equal(km.labels_, again.labels_) and km.inertia_ == again.inertia_
            for key, v in (("n", X.shape[0]), ("d", X.shape[1]), ("k", k), ("x_crc", crc(X, np.float32)),
                           ("n_iter", km.n_iter_), ("inertia", np.float64(km.inertia_)), ("labels_crc", crc(km.labels_, np.int32)),
                           ("centers_crc", crc(km.cluster_centers_, np.float32)),
                           ("sizes_crc", crc(np.bincount(km.labels_, minlength=k), np.int64))):
                rec[key].append(v)
            if i % 20 == 0:
                print(i, X.shape, k, km.n_iter_, flush=True)
    p = os.path.join(HERE, "g31_sklearn_init_sweep.npz")
    np.savez_compressed(p, seeds=np.asarray(seeds, dtype=np.int64), stack=np.asarray("scikit-learn " + sklearn.__version__),
                        **{key: np.asarray(v, dtype=np.float64 if key == "inertia" else np.int64) for key, v in rec.items()})
    print("wrote", p, os.path.getsize(p), "bytes")


def singleton_case(mirrored):
    rs = np.random.RandomState(36)
    others = (np.array([5.0, 5.0, 5.0, 5.0]) + rs.randint(-300, 301, size=(50, 4)) / 1024.0).astype(np.float32)
    lone = np.array([[-5.0, 0.0, 0.0, 0.0]], dtype=np.float32)
    X = np.ascontiguousarray(np.concatenate([lone, others]))
    C0 = np.array([[-8.0, 0.0, 0.0, 0.0], [100.0, 100.0, 100.0, 100.0], [5.0, 5.0, 5.0, 5.0]], dtype=np.float32)
    if mirrored:  # the heaviest cluster gets index 0, the lone point's cluster index 2
        C0 = np.ascontiguousarray(C0[::-1])
    return X, C0


def main_singleton():
    import warnings
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    out = {}
    with threadpool_limits(limits=1), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, mirrored in (("a", False), ("b", True)):
            X, C0 = singleton_case(mirrored)
            out[f"{tag}_X"], out[f"{tag}_C0"] = X, C0
            for it in (1, 2, 3):
                km = KMeans(n_clusters=3, init=C0.copy(), n_init=1, max_iter=it).fit(X)
                out[f"{tag}_centers{it}"] = km.cluster_centers_.astype(np.float32)
                out[f"{tag}_labels{it}"] = km.labels_.astype(np.int32)
                out[f"{tag}_n_iter{it}"] = km.n_iter_
                out[f"{tag}_inertia{it}"] = np.float64(km.inertia_)
            print(tag, "after one iteration:", out[f"{tag}_centers1"].tolist(), out[f"{tag}_labels1"][:4])
    p = os.path.join(HERE, "g36_relocation_takes_only_member.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


def draw_kpp_case(seed):
    """(X centred float32 [n, d], K, random_state) of G37's case ``seed``."""
    rs = np.random.RandomState(seed)
    k = int(rs.choice([410, 700, 1100, 1250])) if rs.rand() < 0.2 else int(rs.randint(1, 61))
    n = int(rs.randint(k + 10, max(k + 11, 6000)))
    d = int(rs.choice([64, 96, 128, 96, int(rs.randint(1, 41)), 4 * int(rs.randint(1, 60))]))
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, 40, 1500, 700)
    if rs.rand() < 0.5:
        X = X * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(X.shape) * 1e-3).astype(np.float32)
    X = X - X.mean(axis=0)
    return np.ascontiguousarray(X.astype(np.float32)), k, int(rs.randint(1 << 30))


def main_kpp(count, seed0):
    from sklearn.cluster import kmeans_plusplus
    from threadpoolctl import threadpool_limits
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    out = dict(seeds=np.asarray(seeds, dtype=np.int64), x_crc=[], k=[])
    with threadpool_limits(limits=1):
        for i, s in enumerate(seeds):
            X, k, state = draw_kpp_case(s)
            _, idx = kmeans_plusplus(X, k, random_state=state)
            _, again = kmeans_plusplus(X, k, random_state=state)
            assert np.array_equal(idx, again)
            out[f"idx{i}"] = np.asarray(idx, dtype=np.int32)
            out["x_crc"].append(crc(X, np.float32))
            out["k"].append(k)
            if i % 20 == 0:
                print(i, X.shape, k, flush=True)
    out["x_crc"], out["k"] = np.asarray(out["x_crc"], dtype=np.int64), np.asarray(out["k"], dtype=np.int64)
    p = os.path.join(HERE, "g37_kmeanspp_sweep.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


def draw_multi_case(seed):
    """(X, C0) of G38's case ``seed``: 2 ... K-1 initial centres out of reach."""
    rs = np.random.RandomState(seed)
    n = int(rs.randint(200, 6000))
    d = int(rs.choice([64, 96, 128, 96, int(rs.randint(2, 41)), 4 * int(rs.randint(1, 60))]))
    k = int(min(rs.randint(4, 80), n // 4))
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, k // 2), 1500, 700)
    if rs.rand() < 0.5:
        X = X * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(X.shape) * 1e-3).astype(np.float32)
    X = np.ascontiguousarray(X.astype(np.float32))
    C0 = X[rs.choice(n, size=k, replace=False)].copy()
    C0[rs.choice(k, size=int(rs.randint(2, k)), replace=False)] += np.float32(50.0)
    return X, C0


def canonical(labels, centers):
    """(labels renamed in order of first appearance, sizes sorted, centres sorted as rows): what is left of a result
    when the names of the clusters do not count."""
    labels = np.asarray(labels)
    _, first = np.unique(labels, return_index=True)
    rename = np.full(int(labels.max()) + 1, -1, dtype=np.int64)
    rename[labels[np.sort(first)]] = np.arange(len(first))
    c = np.ascontiguousarray(centers, dtype=np.float32)
    return rename[labels].astype(np.int32), np.sort(np.bincount(labels)).astype(np.int64), c[np.lexsort(c.T[::-1])]


def main_multi(count, seed0):
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    rec = {key: [] for key in ("n", "d", "k", "x_crc", "n_iter", "inertia", "labels_crc", "centers_crc", "sizes_crc")}
    with threadpool_limits(limits=1):
        for i, s in enumerate(seeds):
            X, C0 = draw_multi_case(s)
            k = C0.shape[0]
            km = KMeans(n_clusters=k, init=C0.copy(), n_init=1).fit(X)
            lab, sizes, cen = canonical(km.labels_, km.cluster_centers_)
            for key, v in (("n", X.shape[0]), ("d", X.shape[1]), ("k", k), ("x_crc", crc(X, np.float32) ^ crc(C0, np.float32)),
                           ("n_iter", km.n_iter_), ("inertia", np.float64(km.inertia_)), ("labels_crc", crc(lab, np.int32)),
                           ("centers_crc", crc(cen, np.float32)), ("sizes_crc", crc(sizes, np.int64))):
                rec[key].append(v)
            if i % 20 == 0:
                print(i, X.shape, k, km.n_iter_, flush=True)
    p = os.path.join(HERE, "g38_sklearn_multi_relocation_sweep.npz")
    np.savez_compressed(p, seeds=np.asarray(seeds, dtype=np.int64),
                        **{key: np.asarray(v, dtype=np.float64 if key == "inertia" else np.int64) for key, v in rec.items()})
    print("wrote", p, os.path.getsize(p), "bytes")


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "multi":
        return main_multi(int(sys.argv[1]), int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "kpp":
        return main_kpp(int(sys.argv[1]), int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "singleton":
        return main_singleton()
    if len(sys.argv) > 3 and sys.argv[3] == "init":
        return main_init(int(sys.argv[1]), int(sys.argv[2]))
    import sklearn
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_info, threadpool_limits
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    blas = [i for i in threadpool_info() if i.get("user_api") == "blas"]
    stack = "numpy %s; scikit-learn %s; %s" % (np.__version__, sklearn.__version__, "; ".join(
        "%s %s core=%s" % (i.get("internal_api"), i.get("version"), i.get("architecture")) for i in blas))
    rec = {key: [] for key in ("n", "d", "k", "n_init", "x_crc", "n_iter", "inertia", "labels_crc", "centers_crc", "sizes_crc")}
    with threadpool_limits(limits=1):
        for i, s in enumerate(seeds):
            X, k, n_init, state = draw_case(s)
            km = KMeans(n_clusters=k, n_init=n_init, random_state=state).fit(X)
            again = KMeans(n_clusters=k, n_init=n_init, random_state=state).fit(X)   # deterministic on one thread?
            assert np.array_equal(km.labels_, again.labels_) and km.inertia_ == again.inertia_
            for key, v in (("n", X.shape[0]), ("d", X.shape[1]), ("k", k), ("n_init", n_init), ("x_crc", crc(X, np.float32)),
                           ("n_iter", km.n_iter_), ("inertia", np.float64(km.inertia_)), ("labels_crc", crc(km.labels_, np.int32)),
                           ("centers_crc", crc(km.cluster_centers_, np.float32)),
                           ("sizes_crc", crc(np.bincount(km.labels_, minlength=k), np.int64))):
                rec[key].append(v)
            if i % 20 == 0:
                print(i, X.shape, k, n_init, km.n_iter_, flush=True)
    p = os.path.join(HERE, "g30_sklearn_sweep.npz")
    np.savez_compressed(p, seeds=np.asarray(seeds, dtype=np.int64), stack=np.asarray(stack),
                        **{key: np.asarray(v, dtype=np.float64 if key == "inertia" else np.int64) for key, v in rec.items()})
    print("wrote", p, os.path.getsize(p), "bytes;", stack)


if __name__ == "__main__":
    main()
