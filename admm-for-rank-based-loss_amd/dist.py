"""Row-sharded ADMM over several GPUs: one process per GPU, ``torch.distributed`` for the
collectives (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

The reference is single-process; the sample axis shards like this
(SURVEY 8e):  rank r holds rows [r*nmax, min((r+1)*nmax, n)) of D and the matching slices
of v, m, z, lambda; w, G, sigma are replicated.  Per iteration:
  1. local  m = D w - lambda/rho                                  (engine.phase_m)
  2. rank weights only: the z-step needs the GLOBAL order of m.  Distributed form (default
     when the engine has the zd_* methods): splitter-based sample sort so that rank r owns a
     contiguous range of the sorted order, exact PAV on the own chunk, then a merge tree over
     ranks whose seams are resolved by a K-ary search that exchanges only (count, sum sigma,
     sum m) summaries; z goes back to the rows' owners (``_z_distributed``).  Replicated form
     (``dist_z=False``): all-gather m (8n bytes), z-step on the gathered vector on every rank.
     erm needs no exchange: z_i = prox(m_i) is local.
  3. local  q = D^T (z + lambda/rho), all-reduce(sum) of q (d doubles)   (phase_q)
  4. replicated d-space w-step                                     (engine.phase_w)
  5. local  v = D w, lambda update, partial sums; all-reduce(sum) of 2 doubles
                                                                   (phase_dual/finish)
Host waits and collectives are counted per iteration (``collectives`` / ``driver_syncs`` on the returned
statistics, next to the library's own ``host_syncs``): the distributed z-step needs ONE host wait for the
count matrix of its forward exchange (the split sizes of an all-to-all are host arguments; the return trip's
matrix is its transpose), one for the rank bounds the merge tree looks at, and one more per tree level that
really has to pool across a rank boundary - levels whose seams are all in order issue no collective at all.
One-time: all-reduce of the local Gram matrices (d*d doubles) and, for generated data, of
the column sums.  The engine is any object with the phase methods below; ``GpuEngine``
binds them to librbl.so, the CPU tests plug in a NumPy engine built on the oracle.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_rows(n_total, world, rank):
    """Contiguous shards of nmax = ceil(n/world) rows; only the last shards can be short,
    so the concatenation of the padded all-gather buffer is contiguous in [0, n_total)."""
    nmax = (n_total + world - 1) // world
    lo = min(rank * nmax, n_total)
    hi = min(lo + nmax, n_total)
    return lo, hi - lo, nmax


class _DevArray:
    """Zero-copy view of a librbl device buffer for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, count, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class GpuEngine:
    """The librbl solver handle of this rank, exposing its exchange buffers as torch tensors."""

    def __init__(self, solver, device):
        self.s = solver
        self.device = torch.device("cuda", device)
        self.n_local, self.n_total, self.d = solver.n, solver.n_total, solver.d
        self.sorted_path = solver.cfg.weight_function != 0
        self.needs_branch_sum = solver.cfg.weight_function == 6      # EHRM: two scalars summed over the ranks
        self._views = {}
        torch.cuda.set_device(self.device)
        # library kernels and torch collectives are ordered on one stream
        self.s.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def buf(self, which):
        from . import _lib
        ids = dict(m=_lib.BUF_M, q=_lib.BUF_Q, red=_lib.BUF_RED, G=_lib.BUF_G, v=_lib.BUF_V,
                   colstats=_lib.BUF_COLSTATS)
        if which not in self._views:
            ptr, cnt = self.s.buffer(ids[which])
            self._views[which] = (torch.as_tensor(_DevArray(ptr, cnt), device=self.device) if cnt > 0
                                  else torch.empty(0, dtype=torch.float64, device=self.device))
        return self._views[which]

    def new(self, count):
        return torch.empty(int(count), dtype=torch.float64, device=self.device)

    def pending_reduce(self):
        return self.s.pending_reduce()

    def synth_local(self, seed, class_sep, flip_y):
        self.s.synth_local(seed, class_sep, flip_y)

    def synth_finish(self):
        self.s.synth_finish()

    def gram_local(self):
        self.s.gram_local()

    def gram_finish(self):
        self.s.gram_finish()

    def phase_m(self):
        self.s.phase_m()

    def phase_z(self, m_all):
        self.s.phase_z(None if m_all is None else m_all.data_ptr())

    def phase_q(self):
        self.s.phase_q()

    def phase_w(self):
        self.s.phase_w()

    def phase_dual(self, want_objective):
        self.s.phase_dual(want_objective)

    def phase_finish(self):
        return self.s.phase_finish()

    def risk_from_v(self, v_all):
        return self.s.risk_from_v(v_all.data_ptr())

    def sync(self):
        torch.cuda.synchronize(self.device)

    # ---------------------------------------------------------------- distributed z-step
    # thin bindings of rbl_zd_* (include/rbl.h); the tensors are views of library buffers
    def _zdv(self, which, typestr):
        key = ("zd", which)
        if key not in self._views:
            ptr, cnt = self.s.buffer(which)
            self._views[key] = torch.as_tensor(_DevArray(ptr, cnt, typestr), device=self.device)
        return self._views[key]

    def _small(self, lo, hi):
        from . import _lib
        return self._zdv(_lib.BUF_ZD_SMALL, "<f8")[lo:hi]

    def zd_sort_local(self, nsamples):
        self.s.zd_sort_local(nsamples)
        return self._small(0, nsamples)

    def zd_partition(self, splitters):
        """per-destination counts of the local sorted run, left on the device (int64 x world)"""
        from . import _lib
        self._zd_world = splitters.numel() + 1
        self.s.zd_partition(splitters.data_ptr(), self._zd_world)
        return self._zdv(_lib.BUF_ZD_COUNTS, "<i8")[: self._zd_world]

    def zd_sort_losses(self, nsamples):
        self.s.zd_sort_losses(nsamples)
        return self._small(0, nsamples)

    def zd_risk(self, nrecv, sigma_off):
        self.s.zd_risk(nrecv, sigma_off)
        return self._small(264, 265)

    def zd_send_buffers(self):
        from . import _lib
        n = self.n_local
        return self._zdv(_lib.BUF_ZD_SKEYS, "<i8")[:n], self._zdv(_lib.BUF_ZD_SIDS, "<i4")[:n]

    def zd_recv_buffers(self, nrecv):
        from . import _lib
        self._zd_n = int(nrecv)
        return self._zdv(_lib.BUF_ZD_RKEYS, "<i8")[:self._zd_n], self._zdv(_lib.BUF_ZD_RIDS, "<i4")[:self._zd_n]

    def zd_prepare(self, nrecv, sigma_off):
        self.s.zd_prepare(nrecv, sigma_off)
        return self._small(260, 262)

    def zd_pav(self, fvals_total):
        self.s.zd_pav(fvals_total.data_ptr())

    def zd_bounds(self):
        self.s.zd_bounds()
        return self._small(256, 259)

    def zd_seam_setup(self, rank, world, level, bounds_all):
        self._zd_world = world
        self.s.zd_seam_setup(rank, world, level, bounds_all.data_ptr())

    def zd_seam_propose(self, K, cand_all_prev, part_sum_prev):
        self.s.zd_seam_propose(K, 0 if cand_all_prev is None else cand_all_prev.data_ptr(),
                               0 if part_sum_prev is None else part_sum_prev.data_ptr())
        return self._small(320, 320 + K)

    def zd_seam_eval(self, K, cand_all):
        self.s.zd_seam_eval(K, cand_all.data_ptr())
        return self._small(512, 512 + 3 * K * self._zd_world)

    def zd_seam_sums(self, K, cand_all_prev, part_sum_prev, nseams):
        self.s.zd_seam_sums(K, cand_all_prev.data_ptr(), part_sum_prev.data_ptr(), nseams)
        return self._small(12800, 12800 + 3 * nseams)

    def zd_seam_fill(self, sums_total, nseams):
        self.s.zd_seam_fill(sums_total.data_ptr())

    def zd_return_partition(self, nmax, world):
        self.s.zd_return_partition(nmax, world)      # no host wait: the driver knows the counts (transpose)

    def zd_back_send(self):
        from . import _lib
        return self._zdv(_lib.BUF_ZD_BIDS, "<i4")[:self._zd_n], self._zdv(_lib.BUF_ZD_BU, "<f8")[:self._zd_n]

    def zd_back_recv(self, n):
        from . import _lib
        return self._zdv(_lib.BUF_ZD_ZIDS, "<i4")[:int(n)], self._zdv(_lib.BUF_ZD_ZU, "<f8")[:int(n)]

    def zd_scatter(self, n):
        self.s.zd_scatter(n)

    # sort-free z-step for banded rank weights (rbl_zbd_*): the views are what the driver sums / gathers
    def zbd_begin(self):
        return self.s.zbd_begin()

    def zbd_hist(self, p):
        from . import _lib
        self.s.zbd_hist(p)
        return self._zdv(_lib.BUF_ZB_HIST, "<i4")

    def zbd_scan(self, p):
        self.s.zbd_scan(p)

    def zbd_eval(self, k):
        from . import _lib
        self.s.zbd_eval(k)
        return self._zdv(_lib.BUF_ZB_TOT, "<f8")

    def zbd_decide(self, k, last):
        return self.s.zbd_decide(k, last)

    def zbd_root_passes(self):
        return self.s.zbd_root_passes()

    def zbd_gather(self, k):
        from . import _lib
        self.s.zbd_gather(k)
        return self._zdv(_lib.BUF_ZB_PACK, "<f8")

    def zbd_finish(self, k, packs_all, world):
        self.s.zbd_finish(k, packs_all.data_ptr(), world)

    def zbd_apply(self):
        return self.s.zbd_apply()


class ShardedADMM:
    """Drives one engine per rank through the phases with the collectives in between."""

    NS = 64      # regular samples per rank for the splitters
    K = 63       # seam-search candidates per rank and round (64-ary search: 5 rounds decide 16M positions)

    def __init__(self, engine, group=None, dist_z=True, world=None, rank=None):
        """world / rank: only for drivers that bring their own collectives by overriding _allreduce,
        _allgather_rows, _gather_small, _gather_counts and _alltoall (tests/test_gpu_dist.py runs 8
        ranks as threads of one process that way); otherwise they come from torch.distributed."""
        self.e = engine
        self.group = group
        if world is not None:
            self.world, self.rank = int(world), int(rank)
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        _, _, self.nmax = shard_rows(engine.n_total, self.world, self.rank)
        self._gather = None
        self.dist_z = bool(dist_z) and hasattr(engine, "zd_sort_local")
        self._stage = dist.is_initialized() and dist.get_backend(group) == "gloo"
        # a 1-rank group normally skips its (identity) all-reduces; bench.py --sharded-driver sets this
        # to issue them anyway and time the collective's launch path on a 1-GPU box
        self.always_allreduce = False
        self.banded_z = os.environ.get("RBL_NO_ZBAND") != "1"   # sort-free z-step for banded rank weights where it applies
        self.n_coll = 0      # collectives issued in the iteration in flight
        self.n_coll_z = 0    # ... of which by its z-step
        self.zb_clusters = 0 # band edges that can pool (sort-free z-step: 6 + 2 collectives per edge in steady state)
        self.n_sync = 0      # host waits of this driver in the iteration in flight (device -> host reads)

    def _allreduce(self, t):
        if (self.world > 1 or self.always_allreduce) and t.numel() > 0:
            self.n_coll += 1
            self._trace("allreduce", t.numel())
            if self._stage and t.is_cuda:          # gloo moves host memory: stage, like the gathers and the all-to-alls
                h = t.cpu()                        # (gloo handed a device tensor runs copies on streams of its own)
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _allgather_rows(self, local):
        """all-gather of an n_local vector -> n_total vector (valid prefix of a padded buffer)."""
        if self.world == 1:
            return local
        if self._gather is None:
            self._gather = self.e.new(self.world * self.nmax)
            self._pad = self.e.new(self.nmax)
        pad = self._pad
        pad[: local.numel()].copy_(local)
        if local.numel() < self.nmax:
            pad[local.numel():].zero_()
        self.n_coll += 1
        dist.all_gather_into_tensor(self._gather, pad, group=self.group)
        return self._gather[: self.e.n_total]

    # ------------------------------------------------------- distributed z-step (rank weights)
    def _trace(self, *what):
        """RBL_DIST_TRACE=<path prefix>: one line per collective and rank, flushed (debugging aid)"""
        pre = os.environ.get("RBL_DIST_TRACE")
        if pre:
            with open("%s.r%d" % (pre, self.rank), "a") as f:
                f.write(" ".join(str(w) for w in what) + "\n")

    def _gather_small(self, t):
        out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
        self.n_coll += 1
        self._trace("gather", t.numel(), str(t.dtype))
        if self._stage and t.is_cuda:          # gloo moves host memory: stage (tests on one GPU only)
            h = torch.empty(out.numel(), dtype=t.dtype)
            dist.all_gather_into_tensor(h, t.contiguous().cpu(), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def _to_host(self, t):
        """the one place the driver waits for the device"""
        self.n_sync += 1
        a = t.cpu().numpy()
        self._trace("to_host", a.tolist() if a.size <= 64 else a.size)
        return a

    def _gather_counts(self, counts_dev):
        """every rank's per-destination counts (an int64 tensor that never left the device) -> the
        (world, world) matrix on the host, row r = what r sends: one collective, ONE host wait"""
        return self._to_host(self._gather_small(counts_dev)).reshape(self.world, self.world)

    def _alltoall(self, send, send_counts, recv, recv_counts):
        sc, rc = [int(c) for c in send_counts], [int(c) for c in recv_counts]
        self.n_coll += 1
        self._trace("alltoall", send.numel(), sc, recv.numel(), rc)
        if self._stage and send.is_cuda:       # gloo moves host memory: stage (tests on one GPU only)
            r = torch.empty(recv.numel(), dtype=recv.dtype)
            dist.all_to_all_single(r, send.cpu(), rc, sc, group=self.group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, rc, sc, group=self.group)

    @staticmethod
    def _splitters(samples_all, world):
        s = samples_all[~torch.isnan(samples_all)]
        if s.numel() == 0:
            return torch.full((world - 1,), float("inf"), dtype=torch.float64, device=samples_all.device)
        s, _ = torch.sort(s)
        idx = [min(s.numel() - 1, (j + 1) * s.numel() // world) for j in range(world - 1)]
        return s[idx].contiguous()

    @staticmethod
    def _level_has_violation(bounds, world, level):
        """does any seam of this tree level have to pool?  Same test as the device's seam set-up
        (pav.hip: k_zd_seam_setup; oracle/zdist.py: seam_of) on the gathered (first, last, count) triples:
        the last non-empty chunk on the left ends ABOVE the first non-empty chunk on the right begins."""
        half = 1 << (level - 1)
        for k in range((world + 2 * half - 1) // (2 * half)):
            a0 = 2 * k * half
            b0 = a0 + half
            if b0 >= world:
                continue
            b1 = min(b0 + half, world)
            left = [r for r in range(a0, b0) if bounds[r, 2] > 0]
            right = [r for r in range(b0, b1) if bounds[r, 2] > 0]
            if left and right and bounds[left[-1], 1] > bounds[right[0], 0]:
                return True
        return False

    def _z_distributed(self):
        """z-step for rank weights with the sorted order partitioned over the ranks (the CPU
        restatement of every engine call is oracle/zdist.py)."""
        from math import ceil, log2
        e, P, K = self.e, self.world, self.K
        # 1. sample sort: splitters from regular samples, rows to the owners of their key range.  The counts
        # never leave the device until the whole matrix is read with one host wait
        samples_all = self._gather_small(e.zd_sort_local(self.NS))
        cm = self._gather_counts(e.zd_partition(self._splitters(samples_all, P)))
        send_counts, recv_counts = cm[self.rank, :], cm[:, self.rank]
        totals = cm.sum(axis=0)
        nrecv, off = int(totals[self.rank]), int(totals[: self.rank].sum())
        sk, si = e.zd_send_buffers()
        rk, ri = e.zd_recv_buffers(nrecv)
        self._alltoall(sk, send_counts, rk, recv_counts)
        self._alltoall(si, send_counts, ri, recv_counts)
        # 2. exact PAV of the own chunk (EHRM: the branch test sums two scalars over the ranks)
        fv = e.zd_prepare(nrecv, off)
        if e.needs_branch_sum:
            self._allreduce(fv)
        e.zd_pav(fv)
        # 3. merge tree over ranks.  The (first, last, count) triples of all chunks are gathered and looked at
        # on the host: a level none of whose seams is out of order costs nothing, and the triples stay valid
        # until a level really pools
        nseams = (P + 1) // 2
        rounds = 1
        s = int(totals.max())
        while s > K:
            s //= (K + 1)
            rounds += 1
        rounds += 1
        bounds_all = bounds_host = None
        for level in range(1, int(ceil(log2(P))) + 1):
            if bounds_all is None:
                bounds_all = self._gather_small(e.zd_bounds())
                bounds_host = self._to_host(bounds_all).reshape(P, 3)
            if not self._level_has_violation(bounds_host, P, level):
                continue
            e.zd_seam_setup(self.rank, P, level, bounds_all)
            cand_all = part = None
            for _ in range(rounds):
                cand_all = self._gather_small(e.zd_seam_propose(K, cand_all, part))
                part = e.zd_seam_eval(K, cand_all)
                self._allreduce(part)
            sums = e.zd_seam_sums(K, cand_all, part, nseams)
            self._allreduce(sums)
            e.zd_seam_fill(sums, nseams)
            bounds_all = None            # pooled values changed the chunk ends
        # 4. block values back to the owners of the rows: the count matrix of the return trip is the
        # transpose of the forward one (chunk b holds cm[a][b] rows of owner a) - no host wait
        e.zd_return_partition(self.nmax, P)
        n_back = int(cm[self.rank, :].sum())
        bi, bu = e.zd_back_send()
        zi, zu = e.zd_back_recv(n_back)
        self._alltoall(bi, recv_counts, zi, send_counts)
        self._alltoall(bu, recv_counts, zu, send_counts)
        e.zd_scatter(n_back)

    def _z_banded(self):
        """z-step for rank weights that are constant on a few bands (superquantile, aorr, aorr_dc) WITHOUT a sort:
        per select pass one sum of a 12 288-bin integer histogram, per root pass (one in steady state) one sum of 64 doubles, one gather of
        the <= 2048 undecided elements per band edge that can pool - no sample sort, no all-to-all, no merge tree
        (include/rbl.h: rbl_zbd_*; csrc/zband.hip).  Every rank holds the same state throughout, so every rank reads
        the same verdict: False = not applicable / not certified, the caller runs the sort-based z-step."""
        e = self.e
        ok, clusters = e.zbd_begin()
        if not ok:
            return False
        self.zb_clusters = len(clusters)
        for p in range(6):
            self._allreduce(e.zbd_hist(p))
            e.zbd_scan(p)
        passes = e.zbd_root_passes()             # (the library's constant, through the C ABI)
        for k in clusters:
            for r in range(passes):
                self._allreduce(e.zbd_eval(k))
                # the verdict of the pass is a function of the SUMMED totals: every rank reads the same one (one host
                # wait on a pinned word) and stops after the pass that settles - in steady state the first, its
                # candidates sit around a prediction from the last block values: 1 all-reduce per edge instead of 4
                self.n_sync += 1
                if e.zbd_decide(k, r == passes - 1):
                    break
            e.zbd_finish(k, self._gather_small(e.zbd_gather(k)), self.world)
        self.n_sync += 1          # the verdict is read on the host
        return e.zbd_apply() == 0

    def _risk_distributed(self):
        """sum_i sigma_i loss_(i) (objective.py:73-82) with the sorted losses partitioned over the
        ranks: the sample sort of the z-step on the loss keys, a dot product per chunk, one sum."""
        e, P = self.e, self.world
        samples_all = self._gather_small(e.zd_sort_losses(self.NS))
        cm = self._gather_counts(e.zd_partition(self._splitters(samples_all, P)))
        totals = cm.sum(axis=0)
        nrecv, off = int(totals[self.rank]), int(totals[: self.rank].sum())
        sk, _ = e.zd_send_buffers()
        rk, _ = e.zd_recv_buffers(nrecv)
        self._alltoall(sk, cm[self.rank, :], rk, cm[:, self.rank])
        part = e.zd_risk(nrecv, off)
        self._allreduce(part)
        return float(self._to_host(part)[0])

    # ------------------------------------------------------------------- one-time setup
    def setup_synthetic(self, seed=17, class_sep=1.0, flip_y=0.01):
        self.e.synth_local(seed, class_sep, flip_y)
        self._allreduce(self.e.buf("colstats"))
        self.e.synth_finish()

    def setup_gram(self):
        self.e.gram_local()
        self._allreduce(self.e.buf("G"))
        self.e.gram_finish()

    # ------------------------------------------------------------------------ iteration
    def step(self, want_objective=False):
        e = self.e
        self.n_coll = self.n_sync = 0
        self._trace("step")
        e.phase_m()
        if e.sorted_path and self.world > 1 and self.dist_z:
            if not (self.banded_z and hasattr(e, "zbd_begin") and self._z_banded()):
                self._z_distributed()
        else:
            m_all = None
            if e.sorted_path and self.world > 1:
                m_all = self._allgather_rows(e.buf("m"))
            e.phase_z(m_all)
        self.n_coll_z = self.n_coll
        e.phase_q()
        pending = getattr(e, "pending_reduce", None)
        if pending is None:                       # plain engines: q and the residuals are two buffers
            self._allreduce(e.buf("q"))
            e.phase_w()
            e.phase_dual(want_objective)
            self._allreduce(e.buf("red"))
        else:
            # librbl keeps [q | seed | ||z||^2 | primal^2 | loss] in ONE buffer: a single-sweep erm
            # iteration needs one collective (after the pass), the unfused path two slices of it
            x = e.buf("q")
            nred = e.buf("red").numel()
            if pending() & 1:
                self._allreduce(x[: x.numel() - nred])
            e.phase_w()
            e.phase_dual(want_objective)
            m = pending()
            if m == 3:
                self._allreduce(x)
            elif m & 2:
                self._allreduce(x[x.numel() - nred:])
        st = e.phase_finish()
        if want_objective and e.sorted_path and self.world > 1:
            # rank-weighted objective needs the global order of v: gather it (logging only);
            # phase_finish returned the regulariser alone in this case
            if self.dist_z and hasattr(e, "zd_risk"):
                st.objective += self._risk_distributed()
            else:
                st.objective += e.risk_from_v(self._allgather_rows(e.buf("v")))
        # how the iteration talked: collectives issued, host waits of this driver (the library's own are
        # st.host_syncs: the statistics block, the w-step's status word)
        st.collectives, st.driver_syncs, st.z_collectives = self.n_coll, self.n_sync, self.n_coll_z
        return st
