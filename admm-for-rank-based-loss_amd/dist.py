"""Row-sharded ADMM over several GPUs: one process per GPU, ``torch.distributed`` for the
collectives (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

The reference is single-process; the sample axis shards like this
(SURVEY 8e):  rank r holds rows [r*nmax, min((r+1)*nmax, n)) of D and the matching slices
of v, m, z, lambda; w, G, sigma are replicated.  Per iteration:
  1. local  m = D w - lambda/rho                                  (engine.phase_m)
  2. rank weights only: all-gather m (8n bytes) so that every rank can rank its rows
     globally, then the z-step on the gathered vector, keeping the local slice
                                                                   (engine.phase_z)
     erm needs no exchange: z_i = prox(m_i) is local.
  3. local  q = D^T (z + lambda/rho), all-reduce(sum) of q (d doubles)   (phase_q)
  4. replicated d-space w-step                                     (engine.phase_w)
  5. local  v = D w, lambda update, partial sums; all-reduce(sum) of 2 doubles
                                                                   (phase_dual/finish)
One-time: all-reduce of the local Gram matrices (d*d doubles) and, for generated data, of
the column sums.  The engine is any object with the phase methods below; ``GpuEngine``
binds them to librbl.so, the CPU tests plug in a NumPy engine built on the oracle.
"""
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

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class GpuEngine:
    """The librbl solver handle of this rank, exposing its exchange buffers as torch tensors."""

    def __init__(self, solver, device):
        self.s = solver
        self.device = torch.device("cuda", device)
        self.n_local, self.n_total, self.d = solver.n, solver.n_total, solver.d
        self.sorted_path = solver.cfg.weight_function != 0
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


class ShardedADMM:
    """Drives one engine per rank through the phases with the collectives in between."""

    def __init__(self, engine, group=None):
        self.e = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        _, _, self.nmax = shard_rows(engine.n_total, self.world, self.rank)
        self._gather = None

    def _allreduce(self, t):
        if self.world > 1 and t.numel() > 0:
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
        dist.all_gather_into_tensor(self._gather, pad, group=self.group)
        return self._gather[: self.e.n_total]

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
        e.phase_m()
        m_all = None
        if e.sorted_path and self.world > 1:
            m_all = self._allgather_rows(e.buf("m"))
        e.phase_z(m_all)
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
            st.objective += e.risk_from_v(self._allgather_rows(e.buf("v")))
        return st
