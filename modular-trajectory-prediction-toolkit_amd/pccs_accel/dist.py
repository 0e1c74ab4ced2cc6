"""Points sharded over the GPUs of one node: one process per GPU, torch.distributed (RCCL).

The E-step is embarrassingly parallel over points; the M-step needs exactly one exchange per
Lloyd iteration: an all-reduce(sum) of the packed double buffer ``red`` = [K*D sums | K counts |
n_changed | spare] (K=1024, D=128: 1.03 MB).  Every rank then finishes the iteration
identically from the reduced buffer (average, centre shift, convergence flags), so centroids
stay replicated bit-for-bit and no broadcast is needed.  Because the sums are doubles the
result does not depend on how many ranks share the points.  Labels stay sharded.

Empty-cluster relocation (rare) needs the globally farthest points: each rank offers its local
top-m, the candidates are all-gathered and every rank picks the same global top-m.

``ShardedLloyd`` also runs with world_size 1 (no collective), which is what bench.py times at
N=1.  ``group``/``backend`` are whatever the caller initialised; the CPU tests drive the same
class logic over gloo with a NumPy stand-in for the kernels (tests/test_dist_gloo.py).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import (FLAG_N_EMPTY, FLAG_N_ITER, FLAG_STATE, FLAG_WORDS, STATE_RELOCATE, STATE_RUNNING, STATE_STRICT,
                   check, lib, ptr, stream_ptr)

def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_total, rank, world_size, align=1):
    """Contiguous row block of ``rank``: the first n_total % world_size ranks get one extra row.  ``align`` > 1:
    every block but the last starts and ends at a multiple of ``align`` (blocks of ``align`` rows are dealt out
    instead of rows; ``seeding.POT_ALIGN`` for the sharded seeding's OpenBLAS-order potentials)."""
    if align > 1:
        lo, hi = shard_bounds((n_total + align - 1) // align, rank, world_size)
        return min(lo * align, n_total), min(hi * align, n_total)
    base, rem = divmod(n_total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _host_staged(group):
    """gloo cannot be trusted with HIP device tensors (its device path hangs on ROCm for buffers
    beyond a few KB), so under gloo -- rehearsals of the multi-rank driver on one GPU, CPU tests --
    device buffers are staged through the host.  RCCL ("nccl") takes the device buffer as is."""
    return dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None):
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast_from(t, src_rank, group=None):
    """t of group rank ``src_rank`` on every rank (in place)."""
    src = dist.get_global_rank(group, src_rank) if group is not None else src_rank
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def all_gather_cat(t, world_size, group=None):
    """cat of every rank's ``t`` (same shape on all ranks) along dim 0, on t's device."""
    src = t.cpu() if (t.is_cuda and _host_staged(group)) else t
    parts = [torch.empty_like(src) for _ in range(world_size)]
    dist.all_gather(parts, src, group=group)
    return torch.cat(parts).to(t.device)


class DeviceOps:
    """The HIP kernels behind the step-level C-ABI (include/mtp_accel.h)."""

    def __init__(self, device):
        self.device = device
        self._ring = None
        self._ring_pos = 0

    def zeros(self, n, dtype):
        return torch.zeros(n, dtype=dtype, device=self.device)

    def begin(self, C, k, d, staged, c2, red, flags):
        check(lib().mtp_lloyd_begin_f32(ptr(C), k, d, ptr(staged), ptr(c2), ptr(red), ptr(flags), stream_ptr()),
              "mtp_lloyd_begin_f32")

    def assign(self, X, C, staged, c2, k, labels, red, flags):
        n, d = X.shape
        check(lib().mtp_lloyd_assign_f32(ptr(X), n, d, ptr(C), ptr(staged), ptr(c2), k, ptr(labels), ptr(red),
                                         ptr(flags), stream_ptr()), "mtp_lloyd_assign_f32")

    def accumulate(self, X, k, labels, red, flags):
        n, d = X.shape
        check(lib().mtp_lloyd_accumulate_f32(ptr(X), n, d, ptr(labels), k, ptr(red), ptr(flags), stream_ptr()),
              "mtp_lloyd_accumulate_f32")

    def mstep(self, red, C_old, C_new, k, d, tol, max_iter, relocation_done, staged, c2, shift, flags, red_backup):
        check(lib().mtp_lloyd_mstep_f32(ptr(red), ptr(C_old), ptr(C_new), k, d, float(tol), int(max_iter),
                                        int(relocation_done), ptr(staged), ptr(c2), ptr(shift), ptr(flags),
                                        ptr(red_backup), stream_ptr()), "mtp_lloyd_mstep_f32")

    def far_points(self, X, C_old, labels, m, idx_offset):
        n, d = X.shape
        val = torch.empty(m, dtype=torch.float32, device=self.device)
        idx = torch.empty(m, dtype=torch.int64, device=self.device)
        lab = torch.empty(m, dtype=torch.int32, device=self.device)
        rows = torch.empty((m, d), dtype=torch.float32, device=self.device)
        scratch = torch.empty(int(lib().mtp_far_points_scratch_bytes(n)), dtype=torch.uint8, device=self.device)
        check(lib().mtp_far_points_f32(ptr(X), n, d, ptr(C_old), ptr(labels), m, int(idx_offset), ptr(val), ptr(idx),
                                       ptr(lab), ptr(rows), ptr(scratch), stream_ptr()), "mtp_far_points_f32")
        return val, idx, lab, rows

    def apply_relocation(self, red, k, d, m, val, lab, rows):
        check(lib().mtp_apply_relocation_f32(ptr(red), k, d, m, ptr(val), ptr(lab), ptr(rows), stream_ptr()),
              "mtp_apply_relocation_f32")

    def final_assign(self, X, C, staged, c2, k, labels):
        # E-step only, ignoring the state word (labels must match the final centres)
        n, d = X.shape
        running = torch.zeros(FLAG_WORDS, dtype=torch.int32, device=self.device)
        scratch = torch.zeros(k * d + k + 2, dtype=torch.float64, device=self.device)
        check(lib().mtp_lloyd_assign_f32(ptr(X), n, d, ptr(C), ptr(staged), ptr(c2), k, ptr(labels), ptr(scratch),
                                         ptr(running), stream_ptr()), "mtp_lloyd_assign_f32")

    def inertia(self, X, C, labels):
        from . import kernels as K
        return K.inertia(X, C, labels)

    def snapshot(self, flags):
        """Asynchronous copy of the flag words to pinned host memory, in stream order behind the
        kernels enqueued so far; ``wait`` returns them without holding up later launches."""
        if self._ring is None:  # at most two snapshots are alive at a time (the lag is one step)
            self._ring = [(torch.empty(FLAG_WORDS, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
                          for _ in range(4)]
        host, ev = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        host.copy_(flags, non_blocking=True)
        ev.record()
        return host, ev

    def wait(self, snap):
        host, ev = snap
        ev.synchronize()
        return host.numpy().copy()


class ShardedLloyd:
    """sklearn ``_kmeans_single_lloyd`` (cluster/_kmeans.py:624-752) over row-sharded points.

    X_local: this rank's centred rows [n_local, D] (float32).  C0: initial centres [K, D],
    identical on every rank.  ``row_offset``: global index of the first local row.
    """

    def __init__(self, X_local, C0, row_offset=0, group=None, ops=None, force_collectives=False):
        # force_collectives: issue every collective even at world_size 1 (a one-rank RCCL
        # communicator), so that the RCCL code path can be exercised on a one-GPU box
        self.X = X_local
        self.n, self.d = X_local.shape
        self.k = C0.shape[0]
        self.group = group
        self.rank, self.world = world()
        self.collective = self.world > 1 or (force_collectives and dist.is_initialized())
        self.row_offset = row_offset
        self.ops = ops if ops is not None else DeviceOps(X_local.device)
        o = self.ops
        k, d = self.k, self.d
        self.red = o.zeros(k * d + k + 2, torch.float64)
        self.red_backup = o.zeros(k * d + k + 2, torch.float64)  # where a paused M-step parks red
        self.flags = o.zeros(FLAG_WORDS, torch.int32)
        n_staged = int(lib().mtp_staged_centers_floats(k, d)) if ops is None else 0
        self.staged = o.zeros(max(n_staged, 4), torch.float32)
        self.c2 = o.zeros(k, torch.float32)
        self.shift = o.zeros(k, torch.float32)
        self.cbuf = [C0.clone().contiguous(), o.zeros(k * d, torch.float32).view(k, d)]
        self.cur = 0
        self.labels = (o.zeros(self.n, torch.int32) - 1).contiguous()
        o.begin(self.cbuf[0], k, d, self.staged, self.c2, self.red, self.flags)

    # -- one iteration, in the three pieces bench.py times separately --
    def assign(self):
        self.ops.assign(self.X, self.cbuf[self.cur], self.staged, self.c2, self.k, self.labels, self.red, self.flags)

    def accumulate(self):
        self.ops.accumulate(self.X, self.k, self.labels, self.red, self.flags)

    def reduce(self):
        if self.collective:
            all_reduce_sum(self.red, self.group)

    def mstep(self, tol, max_iter, relocation_done=0):
        self.ops.mstep(self.red, self.cbuf[self.cur], self.cbuf[self.cur ^ 1], self.k, self.d, tol, max_iter,
                       relocation_done, self.staged, self.c2, self.shift, self.flags, self.red_backup)

    def step(self, tol, max_iter):
        self.assign()
        self.accumulate()
        self.reduce()
        self.mstep(tol, max_iter)
        self.cur ^= 1

    def read_flags(self):
        return self.flags.cpu().numpy()

    def finish_step(self, tol, max_iter):
        """After assign/accumulate/reduce/mstep of one iteration: read the flags (one host
        sync) and, if the M-step paused on an empty cluster, relocate and redo it.  Returns the
        number of clusters relocated (0 almost always)."""
        f = self.read_flags()
        n_reloc = 0
        if f[FLAG_STATE] == STATE_RELOCATE:
            n_reloc = int(f[FLAG_N_EMPTY])
            self._relocate(n_reloc, self.cur)
            self.flags[FLAG_STATE] = STATE_RUNNING
            self.mstep(tol, max_iter, relocation_done=1)
        self.cur ^= 1
        return n_reloc

    def run_fixed(self, n_steps, tol, max_iter, on_phase=None):
        """``n_steps`` further iterations with the flags read ONE STEP LATE: the snapshot of
        iteration i is looked at after iteration i+1 has been enqueued, so the device never waits
        for the host.  When iteration i paused on an empty cluster, the kernels of the attempt
        enqueued behind it were no-ops (state != RUNNING; its all-reduce added zeros to the parked
        buffer): the relocation and the M-step are redone and that attempt is enqueued again.
        ``on_phase(attempt, phase)`` (phase 0..3 = before assign / accumulate / reduce+mstep /
        after) lets bench.py bracket the kernels with events.  Returns (clusters relocated,
        attempts that were no-ops, attempts made, final state word); stops early only when the
        loop converged."""
        start = int(self.read_flags()[FLAG_N_ITER])
        target = start + n_steps
        enq = start          # iterations done once everything enqueued so far has run
        attempt = 0
        noop = []
        pending = None       # snapshot taken behind the previous attempt
        n_reloc = 0
        state = STATE_RUNNING
        cb = on_phase if on_phase is not None else (lambda a, ph: None)
        while True:
            snap = None
            this_attempt = None
            if enq < target and state == STATE_RUNNING:
                this_attempt = attempt
                cb(attempt, 0)
                self.assign()
                cb(attempt, 1)
                self.accumulate()
                cb(attempt, 2)
                self.reduce()
                self.mstep(tol, max_iter)
                cb(attempt, 3)
                self.cur ^= 1
                enq += 1
                attempt += 1
                snap = self.ops.snapshot(self.flags)
            if pending is not None:
                f = self.ops.wait(pending)
                state = int(f[FLAG_STATE])
                if state == STATE_RELOCATE:
                    done = int(f[FLAG_N_ITER])
                    old = done & 1
                    n_reloc += int(f[FLAG_N_EMPTY])
                    self._relocate(f[FLAG_N_EMPTY], old)
                    self.flags[FLAG_STATE] = STATE_RUNNING
                    self.cur = old
                    self.mstep(tol, max_iter, relocation_done=1)
                    self.cur = old ^ 1
                    enq = done + 1
                    if this_attempt is not None:
                        noop.append(this_attempt)  # ran behind the pause; its snapshot is stale
                    state = STATE_RUNNING
                    # the redone M-step may itself end the loop (max_iter / convergence)
                    pending = self.ops.snapshot(self.flags)
                    continue
                if state != STATE_RUNNING and this_attempt is not None:
                    noop.append(this_attempt)  # enqueued behind the iteration that ended the loop
            pending = snap
            if pending is None:
                break
        return n_reloc, noop, attempt, state

    def _relocate(self, m, old):
        """Global top-m farthest points -> apply to the (replicated) reduced buffer."""
        o = self.ops
        m = int(m)  # every empty cluster is relocated (sklearn _k_means_common.pyx:167-211)
        self.red.copy_(self.red_backup)  # the paused M-step parked the reduced buffer there
        val, idx, lab, rows = o.far_points(self.X, self.cbuf[old], self.labels, m, self.row_offset)
        if self.collective:
            gv = all_gather_cat(val, self.world, self.group)
            gi = all_gather_cat(idx, self.world, self.group)
            gl = all_gather_cat(lab, self.world, self.group)
            gr = all_gather_cat(rows, self.world, self.group)
            v = gv.cpu().numpy()
            i = gi.cpu().numpy()
            # descending distance, ties to the lowest global row index; ranks without rows offer -1
            order = np.lexsort((i, -v))
            order = np.array([q for q in order if i[q] >= 0][:m], dtype=np.int64)
            sel = torch.from_numpy(order).to(val.device)
            val, lab, rows = gv[sel].contiguous(), gl[sel].contiguous(), gr[sel].contiguous()
            if val.numel() < m:  # fewer points than empty clusters: pad with "no candidate"
                pad = m - val.numel()
                val = torch.cat([val, val.new_full((pad,), -1.0)])
                lab = torch.cat([lab, lab.new_full((pad,), -1)])
                rows = torch.cat([rows, rows.new_zeros((pad, self.d))])
        o.apply_relocation(self.red, self.k, self.d, m, val, lab, rows)

    def run(self, max_iter=300, tol_abs=0.0, check_every=1):
        """Returns (labels_local int32, inertia (global), centers [K, D], n_iter)."""
        enq = 0
        while True:
            budget = min(check_every, max_iter - enq)
            for _ in range(budget):
                self.step(tol_abs, max_iter)
                enq += 1
            f = self.read_flags()
            done_iters = int(f[FLAG_N_ITER])
            if f[FLAG_STATE] == STATE_RELOCATE:
                old = done_iters & 1
                self._relocate(f[FLAG_N_EMPTY], old)
                self.flags[FLAG_STATE] = STATE_RUNNING
                self.cur = old
                self.mstep(tol_abs, max_iter, relocation_done=1)
                self.cur = old ^ 1
                enq = done_iters + 1
                if enq < max_iter:
                    continue
                f = self.read_flags()
            if f[FLAG_STATE] != STATE_RUNNING and f[FLAG_STATE] != STATE_RELOCATE:
                break
            if enq >= max_iter:
                break
        n_iter = int(f[FLAG_N_ITER])
        fin = n_iter & 1
        centers = self.cbuf[fin]
        if f[FLAG_STATE] != STATE_STRICT:
            self.ops.final_assign(self.X, centers, self.staged, self.c2, self.k, self.labels)
        inertia = self.ops.inertia(self.X, centers, self.labels)
        if self.collective:
            all_reduce_sum(inertia, self.group)
        return self.labels, float(inertia.item()), centers, n_iter


class RcclComm:
    """An RCCL communicator owned by libmtp_accel.so (``mtp_comm_create``), spanning the ranks of the initialised
    ``torch.distributed`` job: rank 0's unique id travels to the others through one all-reduce of 128 bytes on the
    existing process group, then every rank joins ``ncclCommInitRank`` on its own GPU.  With it the Lloyd loop runs
    inside the library (``lloyd_sharded_native``): the per-iteration all-reduce is issued by the library itself, in
    stream order between its kernels, and the interpreter is out of the loop."""

    def __init__(self, device, group=None):
        import ctypes
        self.rank, self.world = world()
        self.device = device
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_char * 128)()
            check(lib().mtp_comm_unique_id(buf), "mtp_comm_unique_id")
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if dist.is_available() and dist.is_initialized() and self.world > 1:
            t = ident.to(torch.int32)
            if dist.get_backend(group) != "gloo":
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)  # zeros everywhere but on rank 0
            ident = t.cpu().to(torch.uint8)
        raw = (ctypes.c_char * 128).from_buffer_copy(bytes(ident.numpy().tobytes()))
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(lib().mtp_comm_create(self.rank, self.world, raw, ctypes.byref(handle)), "mtp_comm_create")
        self.handle = handle

    def close(self):
        if self.handle is not None:
            with torch.cuda.device(self.device):
                check(lib().mtp_comm_destroy(self.handle), "mtp_comm_destroy")
            self.handle = None


def lloyd_sharded_native(comm, X_local, C0, row_offset, n_total, max_iter=300, tol_abs=0.0, check_every=1,
                         want_phase_ms=False, final_estep=True):
    """sklearn ``_kmeans_single_lloyd`` over row-sharded centred points, the whole loop inside the library
    (``mtp_lloyd_sharded_f32``: RCCL all-reduce of the packed sums between the library's own kernels).
    ``comm`` None: one rank holding all points.
    Returns (labels_local int32, inertia (global), centers [K, D], n_iter[, phase_ms (4 floats: E-step, sums, all-reduce, M-step finish)])."""
    import ctypes
    n, d = X_local.shape
    k = C0.shape[0]
    dev = X_local.device
    C = C0.clone().contiguous()
    labels = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    inert = torch.zeros(1, dtype=torch.float64, device=dev)
    ws_bytes = int(lib().mtp_lloyd_sharded_workspace_bytes(k, d, n, comm.world if comm is not None else 1))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    n_iter = ctypes.c_int(0)
    phase = (ctypes.c_float * 4)() if want_phase_ms else None
    with torch.cuda.device(dev):
        check(lib().mtp_lloyd_sharded_f32(comm.handle if comm is not None else None, ptr(X_local) if n else None, n, int(row_offset), int(n_total), d, ptr(C),
                                          k, int(max_iter), float(tol_abs), int(check_every), ptr(labels) if n else None,
                                          ptr(inert), ctypes.byref(n_iter), phase, 0 if final_estep else _lib.LLOYD_NO_FINAL_ESTEP,
                                          ptr(ws), ws_bytes, stream_ptr()), "mtp_lloyd_sharded_f32")
    out = (labels, float(inert.item()), C, int(n_iter.value))
    return out + ([float(v) for v in phase],) if want_phase_ms else out


def is_same_clustering_sharded(l1, l2, k, collective, group=None):
    """sklearn cluster/_k_means_common.pyx:314-328 over sharded labels: l1 -> l2 is a function iff, for every
    label of l1, the smallest and the largest l2 seen with it -- over all ranks -- coincide."""
    dev = l1.device
    mn = torch.full((k,), k, dtype=torch.int64, device=dev)
    mx = torch.full((k,), -1, dtype=torch.int64, device=dev)
    if l1.numel():
        a, b = l1.to(torch.int64), l2.to(torch.int64)
        mn.scatter_reduce_(0, a, b, reduce="amin", include_self=True)
        mx.scatter_reduce_(0, a, b, reduce="amax", include_self=True)
    if collective:
        for t, op in ((mn, dist.ReduceOp.MIN), (mx, dist.ReduceOp.MAX)):
            if t.is_cuda and _host_staged(group):
                h = t.cpu()
                dist.all_reduce(h, op=op, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op, group=group)
    seen = mx >= 0
    return bool((mn[seen] == mx[seen]).all())


def kmeans_fit_sharded(X_local, n_total, C0=None, row_offset=0, tol=1e-4, max_iter=300, check_every=1, group=None,
                       force_collectives=False, lagged_steps=0, n_clusters=None, n_init=20, random_state=None,
                       lloyd_ops=None, seed_ops=None, stats=None, comm=None, stats_order=None, seqsum=None,
                       pot_order=None):
    """sklearn ``KMeans.fit`` over row-sharded, un-centred points (one process per GPU): global mean /
    tolerance by all-reduce of column sums, then

      * ``C0`` given (un-centred [K, D], identical on every rank): ``KMeans(init=C0, n_init=1)``;
      * ``C0`` None: ``KMeans(n_clusters, n_init=n_init)`` with k-means++ seeding -- all restarts are seeded in
        lock step (``seeding.kmeans_plusplus_sharded``), each runs ``ShardedLloyd``, and the best is kept by
        sklearn's rule (cluster/_kmeans.py:1525-1532).  ``random_state`` must denote the SAME stream on every
        rank: an int, or None with ``np.random.seed(s)`` called identically in every process.

    ``stats_order``: "numpy" (default with the HIP kernels) -- the column sums behind ``X.mean(axis=0)`` and sklearn's
    tolerance are float32 chains over ALL rows in row order, as NumPy forms them: rank r continues the chains where
    rank r - 1 stopped (``mtp_col_seqsum_carry_f32``, one small broadcast per rank and pass), so the centred points and
    the tolerance are those of the one-GPU fit and of sklearn, bit for bit; "f64": double sums, one all-reduce each
    (default when the ``stats`` test hooks are given).  ``seqsum``: test hook for the "numpy" order.
    ``pot_order``: the order of the seeding's float32 potentials (``seeding.kmeans_plusplus_sharded``): "openblas"
    (default when every rank's first row is a multiple of 4096, ``shard_bounds(..., align=seeding.POT_ALIGN)``) picks
    sklearn's seeds bit for bit; "f64" works for any split.
    ``comm``: an ``RcclComm``; the Lloyd runs then go through ``mtp_lloyd_sharded_f32`` (loop and all-reduce inside
    the library) instead of the step-level driver ``ShardedLloyd``.
    Returns dict(labels (local), centers, inertia, n_iter, tol_abs).  ``lloyd_ops`` / ``seed_ops`` / ``stats``
    (precomputed (col_sum, col_sqdev) functions) are test hooks standing in for the HIP kernels."""
    from . import seeding
    rank, ws = world()
    collective = ws > 1 or (force_collectives and dist.is_initialized())
    if collective:  # decided from gathered sizes, so that every rank raises (a rank that stopped alone would hang the others)
        sizes = all_gather_cat(torch.tensor([X_local.shape[0]], dtype=torch.int64, device=X_local.device), ws, group)
        if int(sizes.min()) == 0:
            raise ValueError(f"kmeans_fit_sharded: every rank must hold at least one row, got {sizes.tolist()} "
                             "(shard_bounds(..., align=a) leaves ranks empty when n_total < world_size * a)")
    if stats is None:
        from . import kernels as K
        col_sums, col_sqdev, sub_rowvec_ = K.col_sums, K.col_sqdev, K.sub_rowvec_
        if seqsum is None:
            seqsum = K.col_seqsum
    else:
        col_sums, col_sqdev, sub_rowvec_ = stats
    if stats_order is None:
        stats_order = "numpy" if (stats is None or seqsum is not None) else "f64"
    if stats_order == "numpy":
        def chain(center):
            # the ranks take turns in rank order; everyone ends with the sums over all rows
            acc = torch.zeros(X_local.shape[1], dtype=torch.float32, device=X_local.device)
            for r in range(ws if collective else 1):
                if r == rank or not collective:
                    acc = seqsum(X_local, center, acc)
                if collective:
                    broadcast_from(acc, r, group)
            return acc
        s = chain(None).cpu().numpy()
        mean = np.true_divide(s, np.intp(n_total), out=s, casting="unsafe")  # numpy/_core/_methods.py _mean
        mean32 = torch.from_numpy(mean).to(X_local.device)
        tol_abs = 0.0
        if tol != 0:
            sq = chain(mean32).cpu().numpy()
            var = np.true_divide(sq, np.intp(n_total), out=sq, casting="unsafe")  # _var
            tol_abs = float(np.mean(var) * tol)
    elif stats_order == "f64":
        col_sum = col_sums(X_local)
        if collective:
            all_reduce_sum(col_sum, group)
        mean64 = col_sum / n_total
        sqdev = col_sqdev(X_local, mean64)
        if collective:
            all_reduce_sum(sqdev, group)
        tol_abs = float((sqdev / n_total).mean().item()) * tol if tol != 0 else 0.0
        mean32 = mean64.to(torch.float32)
    else:
        raise ValueError(f"stats_order must be 'numpy' or 'f64', got {stats_order!r}")
    Xc = X_local.clone()
    sub_rowvec_(Xc, mean32)

    def lloyd(c0c):
        if comm is not None:  # the whole loop inside the library, its own RCCL communicator
            return lloyd_sharded_native(comm, Xc, c0c, row_offset, n_total, max_iter=max_iter, tol_abs=tol_abs,
                                        check_every=check_every)
        sl = ShardedLloyd(Xc, c0c, row_offset=row_offset, group=group, force_collectives=force_collectives, ops=lloyd_ops)
        if lagged_steps:  # the first iterations with the flags read one step late (ShardedLloyd.run_fixed)
            sl.run_fixed(min(lagged_steps, max_iter), tol_abs, max_iter)
        return sl.run(max_iter=max_iter, tol_abs=tol_abs, check_every=check_every)

    if C0 is not None:
        C0c = C0.clone().contiguous()
        sub_rowvec_(C0c, mean32)
        labels, inertia, centers, n_iter = lloyd(C0c)
        return dict(labels=labels, centers=centers + mean32, inertia=inertia, n_iter=n_iter, tol_abs=tol_abs)

    if n_clusters is None:
        raise ValueError("kmeans_fit_sharded: give initial centres C0 or n_clusters")
    if n_total < n_clusters:
        raise ValueError(f"n_samples={n_total} should be >= n_clusters={n_clusters}.")
    if random_state is None:
        rs = np.random.mtrand._rand
    elif isinstance(random_state, (int, np.integer)):
        rs = np.random.RandomState(random_state)
    else:
        rs = random_state
    # the draws of every restart, in sklearn's order (seeding consumes the stream, Lloyd does not)
    draws = [seeding.draw(rs, n_total, n_clusters) for _ in range(n_init)]
    c0s, _ = seeding.kmeans_plusplus_sharded(Xc, row_offset, n_total, n_clusters, draws, group=group, ops=seed_ops,
                                             force_collectives=force_collectives, pot_order=pot_order)
    best = None
    for r in range(n_init):
        labels, inertia, centers, n_iter = lloyd(c0s[r].clone())  # a copy: row r starts 16-byte aligned only when 4 | K * D
        if best is None or (inertia < best["inertia"] and not is_same_clustering_sharded(
                labels, best["labels"], n_clusters, collective, group)):
            best = dict(labels=labels.clone(), centers=centers.clone(), inertia=inertia, n_iter=n_iter)
    best["centers"] = best["centers"] + mean32
    best["tol_abs"] = tol_abs
    return best


def best_of_k_sharded(pred_local, gt_local, n_total, mode=0, ade_div=None, best_of=None, group=None,
                      force_collectives=False, kernel=None):
    """Best-of-K ADE/FDE (PCCSNet/train_steps.py:214-233, StimulusVerification/verify_predictions.py:238-248)
    with the agents sharded over the ranks: every rank scores its own agents (no exchange), then ONE
    all-reduce of the two double sums gives the mean over all ``n_total`` agents on every rank.

    Returns the local per-agent results of ``kernels.best_of_k`` plus ``mean_ade`` / ``mean_fde``
    (Python floats, global).  ``kernel``: test hook, a stand-in for ``kernels.best_of_k``."""
    if kernel is None:
        from . import kernels as K
        kernel = K.best_of_k
    rank, ws = world()
    r = kernel(pred_local, gt_local, mode=mode, ade_div=ade_div, best_of=best_of)
    sums = r["sums"]
    if ws > 1 or (force_collectives and dist.is_initialized()):
        all_reduce_sum(sums, group)
    total = sums.cpu()
    r["mean_ade"] = float(total[0]) / n_total
    r["mean_fde"] = float(total[1]) / n_total
    return r
