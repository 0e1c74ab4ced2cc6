"""k-means++ seeding on the device, replaying scikit-learn's draws.

Restates sklearn cluster/_kmeans.py:174-272 (``_kmeans_plusplus``, unit sample weights) as
KMeans.fit calls it for every one of the ``n_init`` restarts (PCCSNet/utils/utility.py:120-121
hard-codes n_init=20).  The random numbers come from the same NumPy ``RandomState`` in the same
order (one ``choice`` for the first centre, then ``uniform(size=2+int(ln K))`` per further
centre) and are uploaded once; everything that touches the N points runs in the HIP kernels of
csrc/seeding.hip (``mtp_kmeans_plusplus_f32``) without a host sync: candidate ids, potentials
and the closest-distance rows never leave the device.

Distances follow sklearn.metrics.pairwise._euclidean_distances for float32 input
(pairwise.py:391-442, :582-653): float64 arithmetic (-2 x.y + ||x||^2 + ||y||^2), float32
storage, clamped at 0.  The float32 potentials -- sklearn's are BLAS sums (``rows @ sample_weight``), and
the draws are ``u * pot`` -- are summed in the float addition order of the reference stack (NumPy +
OpenBLAS 0.3.29 x86-64 kernels: ``pot_order="openblas"``, the default), which makes the seeds sklearn's
bit for bit at the reference's real sizes (fixture G14); ``pot_order="f64"`` sums in double and rounds
(then a draw within ~1e-7 relative of a cumulative-sum boundary can pick the neighbouring point: about
one restart in four gets other seeds at N ~ 7e4).  The sharded seeding uses double sums.
"""
import numpy as np
import torch

from ._lib import POT_F64, POT_OPENBLAS, check, lib, ptr, stream_ptr

POT_ORDERS = {"openblas": POT_OPENBLAS, "f64": POT_F64}


def _pot_order(name):
    if name not in POT_ORDERS:
        raise ValueError(f"pot_order must be one of {sorted(POT_ORDERS)}, got {name!r}")
    return POT_ORDERS[name]


def kmeans_plusplus(Xc, n_clusters, random_state, n_local_trials=None, pot_order="openblas"):
    """Returns (centers float32 [K, D] device tensor, indices int64 [K] device tensor)."""
    if not (isinstance(Xc, torch.Tensor) and Xc.is_cuda and Xc.dtype == torch.float32 and Xc.dim() == 2
            and Xc.is_contiguous()):
        raise TypeError("kmeans_plusplus expects a contiguous float32 HIP device tensor [N, D]")
    n, d = Xc.shape
    dev = Xc.device
    if n_local_trials is None:
        n_local_trials = 2 + int(np.log(n_clusters))
    center_id, u = draw(random_state, n, n_clusters, n_local_trials)
    u_dev = torch.from_numpy(np.ascontiguousarray(u.reshape(-1))).to(dev) if u.size else None
    indices = torch.empty(n_clusters, dtype=torch.int64, device=dev)
    centers = torch.empty((n_clusters, d), dtype=torch.float32, device=dev)
    ws_bytes = int(lib().mtp_kpp_workspace_bytes(n, d, n_local_trials))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().mtp_kmeans_plusplus_f32(ptr(Xc), n, d, n_clusters, n_local_trials, center_id, ptr(u_dev),
                                            _pot_order(pot_order), ptr(indices), ptr(centers), ptr(ws), ws_bytes,
                                            stream_ptr()),
              "mtp_kmeans_plusplus_f32")
    return centers, indices


def blas_order_rowsums(rows):
    """float32 [L, n] device rows -> float32 [L]: ``rows @ ones`` in the reference stack's float order
    (see ``mtp_blas_order_rowsums_f32``); L = 1 follows sdot, L >= 2 sgemv_t."""
    if not (isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 2
            and rows.is_contiguous()):
        raise TypeError("blas_order_rowsums expects a contiguous float32 HIP device tensor [L, n]")
    l, n = rows.shape
    out = torch.empty(l, dtype=torch.float32, device=rows.device)
    ws_bytes = max(int(lib().mtp_blas_order_rowsums_workspace_bytes(n, l)), 256)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=rows.device)
    with torch.cuda.device(rows.device):
        check(lib().mtp_blas_order_rowsums_f32(ptr(rows), l, n, ptr(out), ptr(ws), ws_bytes, stream_ptr()),
              "mtp_blas_order_rowsums_f32")
    return out


_CDF = {}


def _uniform_cdf(n):
    """The cdf ``RandomState.choice(n, p=w / w.sum())`` builds for unit float32 weights
    (numpy/random/mtrand.pyx: p -> float64, cdf = p.cumsum(); cdf /= cdf[-1]); it only depends on n."""
    if n not in _CDF:
        _CDF.clear()
        w = np.ones(n, dtype=np.float32)
        cdf = (w / w.sum()).astype(np.float64).cumsum()
        cdf /= cdf[-1]
        _CDF[n] = cdf
    return _CDF[n]


def draw(random_state, n, n_clusters, n_local_trials=None):
    """The random numbers one ``_kmeans_plusplus`` call consumes, in sklearn's order:
    (first centre index, uniform [K-1, L]).  ``choice(n, p=...)`` is one ``random_sample()``
    looked up in the cdf (side='right'); done that way the O(n) cdf is built once, not per restart."""
    if n_local_trials is None:
        n_local_trials = 2 + int(np.log(n_clusters))
    center_id = int(_uniform_cdf(n).searchsorted(random_state.random_sample(), side="right"))
    u = random_state.uniform(size=(max(n_clusters - 1, 0), n_local_trials))
    return center_id, u


def kmeans_plusplus_batched(Xc, n_clusters, draws, pot_order="openblas"):
    """Seeding of len(draws) restarts in shared launches.  ``draws``: list of ``draw()`` results,
    one per restart, taken from the RandomState in restart order.  Returns (centers float32
    [B, K, D], indices int64 [B, K]) device tensors."""
    if not (isinstance(Xc, torch.Tensor) and Xc.is_cuda and Xc.dtype == torch.float32 and Xc.dim() == 2
            and Xc.is_contiguous()):
        raise TypeError("kmeans_plusplus_batched expects a contiguous float32 HIP device tensor [N, D]")
    n, d = Xc.shape
    dev = Xc.device
    b = len(draws)
    l = draws[0][1].shape[1] if n_clusters > 1 else 2 + int(np.log(n_clusters))
    ids = np.asarray([c for c, _ in draws], dtype=np.int64)
    if ids.min() < 0 or ids.max() >= n:
        raise ValueError("first-centre index out of range")
    if n_clusters > 1 and any(tuple(u.shape) != (n_clusters - 1, l) for _, u in draws):
        # the kernel strides through the draws as [B, K-1, L]: anything else would be mis-indexed silently
        raise ValueError(f"every restart's uniform draws must have shape ({n_clusters - 1}, {l})")
    ids_dev = torch.from_numpy(ids).to(dev)
    u_dev = None
    if n_clusters > 1:
        u = np.ascontiguousarray(np.stack([u for _, u in draws]).reshape(-1))
        u_dev = torch.from_numpy(u).to(dev)
    indices = torch.empty((b, n_clusters), dtype=torch.int64, device=dev)
    centers = torch.empty((b, n_clusters, d), dtype=torch.float32, device=dev)
    ws_bytes = int(lib().mtp_kpp_batched_workspace_bytes(n, d, l, b))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().mtp_kmeans_plusplus_batched_f32(ptr(Xc), n, d, n_clusters, l, b, ptr(ids_dev), ptr(u_dev),
                                                    _pot_order(pot_order), ptr(indices), ptr(centers), ptr(ws), ws_bytes,
                                                    stream_ptr()),
              "mtp_kmeans_plusplus_batched_f32")
    return centers, indices


# ------------------------------------------------------------------------------------------
# Row-sharded points: one process per GPU
# ------------------------------------------------------------------------------------------
class DeviceSeedOps:
    """The HIP kernels behind the sharded seeding (include/mtp_accel.h, step-level seeding entries)."""

    def row_sqnorm(self, X):
        n, d = X.shape
        yy = torch.empty(max(n, 1), dtype=torch.float64, device=X.device)
        with torch.cuda.device(X.device):
            check(lib().mtp_kpp_row_sqnorm_f64(ptr(X), n, d, ptr(yy), stream_ptr()), "mtp_kpp_row_sqnorm_f64")
        return yy

    def candidate_dists(self, X, yy, cand, closest):
        """cand float32 [B, L, D] -> (rows float32 [B, L, n] = min(closest, d(cand, x)), local potentials float64 [B, L])."""
        n, d = X.shape
        b, l, _ = cand.shape
        dev = X.device
        rows = torch.empty((b, l, n), dtype=torch.float32, device=dev)
        pot = torch.empty((b, l), dtype=torch.float64, device=dev)
        ws_bytes = max(int(lib().mtp_kpp_candidate_dists_workspace_bytes(max(n, 1), d, b)), 256)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(lib().mtp_kpp_candidate_dists_f32(ptr(X), n, d, ptr(yy), ptr(cand), b, l, ptr(closest), ptr(rows), ptr(pot),
                                                    ptr(ws), ws_bytes, stream_ptr()), "mtp_kpp_candidate_dists_f32")
        return rows, pot


    def block_sums(self, rows, pot64=None, eps=0.0):
        """rows float32 [B, L, n] -> float32 [B, L, nb + 1]: OpenBLAS sgemv_t's 4096-point block sums of every row and,
        last, the sum of its n & 3 trailing points (``mtp_blas_order_block_sums_f32``).  ``pot64`` float64 [B, L]
        (the double sums of the whole rows) and ``eps``: rows that cannot hold the smallest float32 sum are skipped,
        their block sums are zero."""
        b, l, n = rows.shape
        nb = int(lib().mtp_blas_order_blocks(n))
        part = (torch.empty if pot64 is None else torch.zeros)((b, l, nb + 1), dtype=torch.float32, device=rows.device)
        if pot64 is not None:
            pot64 = pot64.contiguous()
        with torch.cuda.device(rows.device):
            check(lib().mtp_blas_order_block_sums_f32(ptr(rows) if n else None, l, n, b, ptr(pot64), float(eps), ptr(part),
                                                      stream_ptr()), "mtp_blas_order_block_sums_f32")
        return part

    def combine(self, parts):
        """float32 [R, m] -> [R]: the entries of a row added one after the other in float (``mtp_blas_order_combine_f32``)."""
        parts = parts.contiguous()
        out = torch.empty(parts.shape[0], dtype=torch.float32, device=parts.device)
        with torch.cuda.device(parts.device):
            check(lib().mtp_blas_order_combine_f32(ptr(parts), parts.shape[1], parts.shape[0], ptr(out), stream_ptr()),
                  "mtp_blas_order_combine_f32")
        return out

    def sdot_carry(self, x, main_local, left_over, acc, finish):
        """One rank's piece of OpenBLAS' sdot over rows x float32 [B, n_local]: acc float32 [B, 64] is continued in
        place over the first main_local points; with ``finish`` the left_over points behind them are taken in and the
        sums float32 [B] returned (``mtp_blas_order_sdot_carry_f32``)."""
        b = acc.shape[0]
        out = torch.empty(b, dtype=torch.float32, device=acc.device) if finish else None
        with torch.cuda.device(acc.device):
            check(lib().mtp_blas_order_sdot_carry_f32(ptr(x) if x.numel() else None, x.stride(0) if x.numel() else 0, b,
                                                      int(main_local), int(left_over), ptr(acc), 1 if finish else 0,
                                                      ptr(out), stream_ptr()), "mtp_blas_order_sdot_carry_f32")
        return out

    def count_below(self, closest, thresholds):
        """counts int64 [B, L]: this rank's cumulative sums of closest[b] (double) below thresholds[b, l]."""
        b, n = closest.shape
        l = thresholds.shape[1]
        dev = closest.device
        counts = torch.empty((b, l), dtype=torch.int64, device=dev)
        ws_bytes = max(int(lib().mtp_kpp_count_below_workspace_bytes(max(n, 1), b)), 256)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        thresholds = thresholds.contiguous()
        with torch.cuda.device(dev):
            check(lib().mtp_kpp_count_below_f32(ptr(closest), n, b, ptr(thresholds), l, ptr(counts), ptr(ws), ws_bytes,
                                                stream_ptr()), "mtp_kpp_count_below_f32")
        return counts


def _first_argmin(v):
    """np.argmin along dim 1: the FIRST minimum."""
    l = v.shape[1]
    m = v.min(dim=1, keepdim=True).values
    pos = torch.arange(l, device=v.device).expand_as(v)
    return torch.where(v == m, pos, torch.full_like(pos, l)).min(dim=1).values


POT_ALIGN = 4096  # OpenBLAS sgemv_t's block length: the shards' first rows must be multiples of it for pot_order="openblas"


def kmeans_plusplus_sharded(X_local, row_offset, n_total, n_clusters, draws, group=None, ops=None,
                            force_collectives=False, packed=True, pot_order=None):
    """k-means++ (sklearn cluster/_kmeans.py:174-272, unit weights) for ``len(draws)`` restarts in lock
    step over ROW-SHARDED centred points: this rank holds rows [row_offset, row_offset + n_local) of
    the n_total (consecutive blocks in rank order).  ``draws``: one ``draw()`` result per restart,
    identical on every rank.

    Two exchanges per new centre (``packed``), both all-gathers of a few KB per rank:
      (A) every rank counts its cumulative sums below each draw (``mtp_kpp_count_below_f32``; the draw is
          taken relative to the rank's first row with the lower ranks' totals, known from (B) of the step
          before; no cumulative sum is materialised) and sends [count | the row it would contribute] --
          its row number ``min(count, n_local - 1)`` -- per (restart, trial).  The sum of the counts IS
          sklearn's ``searchsorted(cumsum, rand)``, it names the owning rank, and the owner's row is
          already there.  (In exact arithmetic the owner's count is the candidate's local row number; if
          rounding between a rank's total and its cumulative sums ever breaks that, every rank sees it in the
          same gathered data, and the seeding is redone with the row fetched by index, ``packed=False``.)
      (B) each rank computes the candidates' distances to its points with the fp64 matrix-pipe kernel; the
          ranks' parts of the [B, L] potentials are gathered and added in rank order: the totals, and every
          rank's part of the chosen row's total for the next step's (A).
    ``pot_order``: "openblas" -- the float32 potentials are added in the reference stack's order (NumPy + OpenBLAS
    sgemv_t / sdot, see ``kmeans_plusplus``): every rank forms the 4096-point block sums of its piece of the candidate
    rows (``mtp_blas_order_block_sums_f32``), a third all-gather (B') brings them together and they are added in rank
    order; the first centre's sdot accumulators are handed from rank to rank once.  Only the rows whose double total
    from (B) can still hold the smallest float32 potential are summed (the one-GPU path's rule, ``kpp_contender`` in
    seeding.hip) -- usually one of the L.  Needs every rank's first row at a multiple of 4096
    (``dist.shard_bounds(..., align=POT_ALIGN)``); then the seeds are sklearn's bit for bit, as on one GPU.
    "f64": double sums (any row split).  Default: "openblas" when the split allows it.
    ``packed=False``: three collectives per centre (counts all-reduced, rows fetched by index, potentials as in (B)).
    Returns (centres float32 [B, K, D], indices int64 [B, K]), replicated.  Needs D % 16 == 0 and at most 8
    local trials (K <= 1096)."""
    from . import dist as pdist
    if ops is None:
        ops = DeviceSeedOps()
    rank, ws = pdist.world()
    collective = ws > 1 or (force_collectives and pdist.dist.is_initialized())
    dev = X_local.device
    n_local, d = X_local.shape
    lo, hi = int(row_offset), int(row_offset) + n_local
    b = len(draws)
    ids0 = torch.tensor([c for c, _ in draws], dtype=torch.int64, device=dev)
    if int(ids0.min()) < 0 or int(ids0.max()) >= n_total:
        raise ValueError("first-centre index out of range")
    if n_clusters > 1 and any(u.shape[0] < n_clusters - 1 or u.shape[1] != draws[0][1].shape[1] for _, u in draws):
        raise ValueError("every restart needs at least n_clusters - 1 rows of uniform draws of one width")
    ar = torch.arange(b, device=dev)

    def gather(t):
        """[W, ...]: every rank's t, in rank order."""
        return pdist.all_gather_cat(t[None].contiguous(), ws, group)

    if collective:
        meta = gather(torch.tensor([lo, n_local], dtype=torch.int64, device=dev))  # [W, 2]
        los, sizes = meta[:, 0].contiguous(), meta[:, 1].contiguous()
        his = (los + sizes).contiguous()
        if int(los[0]) != 0 or int(his[-1]) != n_total or bool((los[1:] != his[:-1]).any()):
            raise ValueError("the ranks must hold consecutive row blocks of the n_total rows, in rank order")

    if collective:
        aligned = bool(((los % POT_ALIGN == 0) | (sizes == 0)).all())  # empty pieces sit at the end
    else:
        los = torch.zeros(1, dtype=torch.int64, device=dev)
        sizes = torch.tensor([n_local], dtype=torch.int64, device=dev)
        aligned = True
    if pot_order is None:
        pot_order = "openblas" if (aligned and hasattr(ops, "block_sums")) else "f64"
    if pot_order not in ("openblas", "f64"):
        raise ValueError(f"pot_order must be 'openblas' or 'f64', got {pot_order!r}")
    if pot_order == "openblas" and not aligned:
        raise ValueError(f"pot_order='openblas' needs every rank's first row at a multiple of {POT_ALIGN}")
    blas = pot_order == "openblas"
    nranks = ws if collective else 1
    sizes_h = [int(v) for v in sizes.cpu()]
    los_h = [int(v) for v in los.cpu()]
    if blas:
        nb_pad = max(int(((m & ~3) + POT_ALIGN - 1) // POT_ALIGN) for m in sizes_h)  # blocks of the longest piece
        # a float32 chain of at most 1100 + (blocks of all pieces) additions of non-negative terms is within that many
        # 2^-24 of the exact sum: a row whose double total exceeds the smallest by more than twice that cannot win
        pot_eps = (1100.0 + nranks * nb_pad) / 8388608.0

    def fetch_rows(idx):
        """X[idx] for GLOBAL row indices idx [B, m] -> [B, m, D] on every rank (owners contribute, one all-reduce)."""
        if n_local:
            # gather with clamped local indices and zero the rows of other ranks: no boolean-mask
            # indexing, which would cost a host synchronisation per call
            own = (idx >= lo) & (idx < hi)
            rows = X_local[(idx - lo).clamp(0, n_local - 1)] * own[..., None].to(torch.float32)
        else:
            rows = torch.zeros(idx.shape + (d,), dtype=torch.float32, device=dev)
        if collective:
            pdist.all_reduce_sum(rows, group)
        return rows

    def total_potentials(pot, rows=None):
        """(global potentials [B, m] in double, every rank's part [W, B, m], float32 potentials in OpenBLAS order or
        None); parts added in rank order.  ``rows`` [B, m, n_local]: this rank's piece of the candidate rows."""
        parts = gather(pot) if collective else pot[None]
        tot = parts[0].clone()
        for r in range(1, parts.shape[0]):
            tot += parts[r]
        pot32 = None
        if blas and rows is not None:
            m = rows.shape[1]
            # rows that can still hold the smallest float32 potential (the same on every rank: from the gathered totals)
            alive = tot * (1.0 - pot_eps) <= tot.min(dim=1, keepdim=True).values * (1.0 + pot_eps)
            bs = ops.block_sums(rows, tot, pot_eps)
            mine = torch.zeros((b, m, nb_pad + 1), dtype=torch.float32, device=dev)  # [block sums, zero padded | tail sum]
            mine[..., :bs.shape[2] - 1] = bs[..., :-1]
            mine[..., -1] = bs[..., -1]
            every = gather(mine) if collective else mine[None]
            # the pieces' block sums side by side in rank order, then added one after the other in float
            seq = every.permute(1, 2, 0, 3).reshape(b * m, -1)
            pot32 = ops.combine(seq).view(b, m)
            pot32 = torch.where(alive, pot32, torch.full_like(pot32, float("inf")))
        return tot, parts, pot32

    def first_potential(closest):
        """[1, n] @ [n] of the first centre's closest row in OpenBLAS' sdot order: the 64 accumulators go from rank to
        rank over the first n_total & -64 points, the rank holding the last point finishes."""
        n64 = (n_total & -32) & -64
        left_over = n_total - n64
        acc = torch.zeros((b, 64), dtype=torch.float32, device=dev)
        out = torch.zeros(b, dtype=torch.float32, device=dev)
        finisher = max(r for r in range(nranks) if sizes_h[r] > 0)
        for r in range(nranks):
            if r == rank or not collective:
                main_local = max(0, min(los_h[r] + sizes_h[r], n64) - los_h[r])
                res = ops.sdot_carry(closest, main_local, left_over, acc, r == finisher)
                if r == finisher:
                    out = res
            if collective:
                pdist.broadcast_from(acc, r, group)
        if collective:
            pdist.broadcast_from(out, finisher, group)
        return out

    yy = ops.row_sqnorm(X_local)
    indices = torch.empty((b, n_clusters), dtype=torch.int64, device=dev)
    rows, pot = ops.candidate_dists(X_local, yy, fetch_rows(ids0[:, None]), None)
    pot, parts, _ = total_potentials(pot)
    lower_tot = parts[:rank, :, 0].sum(dim=0)  # the lower ranks' totals of the closest-distance row
    closest = rows[:, 0, :].contiguous()
    cur_pot = first_potential(closest) if blas else pot[:, 0].to(torch.float32)  # sklearn's potentials are float32
    indices[:, 0] = ids0
    bad = torch.zeros((), dtype=torch.bool, device=dev)
    if n_clusters > 1:
        u = torch.from_numpy(np.ascontiguousarray(np.stack([uu for _, uu in draws]))).to(dev)  # [B, K-1, L]
    for c in range(1, n_clusters):
        rand = u[:, c - 1, :] * cur_pot.to(torch.float64)[:, None]
        if collective:
            rand = rand - lower_tot[:, None]  # the draw relative to this rank's first row
        cnt = ops.count_below(closest, rand)  # cumulative sums < draw, per (restart, trial)
        if not collective:
            idx = cnt.clamp(max=n_total - 1)
            cand = X_local[idx]
        elif packed:
            l = cnt.shape[1]
            payload = torch.zeros((b, l, 1 + d), dtype=torch.float64, device=dev)
            payload[..., 0] = cnt.to(torch.float64)
            if n_local:
                payload[..., 1:] = X_local[cnt.clamp(max=n_local - 1)].to(torch.float64)
            every = gather(payload)  # [W, B, L, 1 + D]
            cnts = every[..., 0].to(torch.int64)
            idx = cnts.sum(dim=0).clamp(max=n_total - 1)
            owner = torch.searchsorted(his, idx, right=True).clamp(max=ws - 1)  # first rank whose block ends beyond idx
            sent = torch.minimum(cnts.gather(0, owner[None])[0], sizes[owner] - 1)  # the row number the owner sent
            bad |= (sent != idx - los[owner]).any()
            cand = every.gather(0, owner[None, :, :, None].expand(1, b, l, 1 + d))[0][..., 1:].to(torch.float32)
        else:
            pdist.all_reduce_sum(cnt, group)
            idx = cnt.clamp(max=n_total - 1)
            cand = fetch_rows(idx)
        rows, pot = ops.candidate_dists(X_local, yy, cand.contiguous(), closest)
        pot, parts, pot32 = total_potentials(pot, rows)
        if pot32 is None:
            pot32 = pot.to(torch.float32)
        best = _first_argmin(pot32)
        closest = rows[ar, best].contiguous()
        cur_pot = pot32[ar, best]
        lower_tot = parts[:rank, ar, best].sum(dim=0)
        indices[:, c] = idx[ar, best]
    if collective and packed and bool(bad):
        return kmeans_plusplus_sharded(X_local, row_offset, n_total, n_clusters, draws, group=group, ops=ops,
                                       force_collectives=force_collectives, packed=False, pot_order=pot_order)
    return fetch_rows(indices), indices
