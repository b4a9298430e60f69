"""Time-step sharding of one transcription across the GPUs of a node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no parallelism of any kind; what is partitioned here is its main loop
`for i in 1:docp.time.steps` (src/DOCP_functions.jl:92-98): step i reads X_i, U_i, K_i, X_{i+1} and v, and writes only
its own rows c[(i-1)(eqs+p)+1 : i(eqs+p)] and the Jacobian entries of those rows.

  * rank r evaluates the contiguous block of steps [r N/G, (r+1) N/G); the first rank also owns the irregular first-step
    columns, the last rank the final-state columns;
  * the ITERATE is sharded like the steps: rank r holds the variables of its own steps inside a
    full-length buffer (global indexing, so the engine's shard handles read it as it is).  What a rank needs from the
    others is tiny: the state X of the NEXT rank's first node (+ its control for trapeze), the first and the final state for
    the boundary rows, and the replicated optimisation variables v.  `exchange_halo` moves them with ONE all-gather of
    n (+m) + n doubles per rank per iterate;
  * a consumer that keeps the whole iterate on one rank calls `broadcast_iterate` instead: one broadcast of nvar doubles;
  * each rank writes its rows straight into a full-length c buffer at their global position; the p+bc tail rows
    (final-time path and boundary constraints) are authoritative on the LAST rank (it owns X_{N+1} and the last controls and
    receives X_1); other ranks only hold them after stitching;
  * outputs stay ROW-SHARDED: rank r holds its rows of c and, for the Jacobian values, one contiguous range of the global
    CSC value array (its step columns) plus its slice of every V column -- what a distributed KKT consumer wants;
    `DOCP.shard` gives the ranges.  The evaluation itself needs no collective;
  * a consumer that wants the residual vector whole on every rank asks for it (`stitch=True`): ONE all-gather of the row
    blocks per evaluation (in place when the blocks are equal, through a padded buffer + one index kernel when ragged).

Every collective is enqueued on torch's current stream, and so are the engine's kernels (`ShardedDOCP` rebinds the handle
to that stream): nothing here synchronises with the host.
"""
import os

import torch
import torch.distributed as dist

# Rehearsal knob: run the collectives on a ONE-rank group too (they are no-ops for the data, but every RCCL call of the
# multi-GPU path -- in-place / padded all-gather, halo all-gather, all-reduce, broadcast -- is issued exactly as on 8 GPUs),
# so the one-GPU box can exercise the real backend
_FORCE = os.environ.get("CTD_DIST_FORCE") == "1"


def shard_steps(N, world, rank):
    """Contiguous block of ceil/floor(N / world) steps for `rank` (equal blocks when world divides N)."""
    base, rem = divmod(N, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


_FLAT_GATHER = {}


def _all_gather_into(recv, send, group=None):
    """recv[r * len(send) : (r+1) * len(send)] = rank r's `send`.  RCCL ("nccl"): one all_gather_into_tensor (send may be a
    view of recv: in place), unguarded -- a failing collective raises on the rank it fails on.  gloo (CPU tests and the
    one-GPU rehearsal; it moves host memory only) goes through the list form, staged on the host when the tensors live on a
    GPU.  The choice is made once per group from the backend's name, never by catching an error of the collective."""
    flat = _FLAT_GATHER.get(group)
    if flat is None:
        flat = _FLAT_GATHER[group] = str(dist.get_backend(group)).lower() in ("nccl", "rccl")
    if flat:
        dist.all_gather_into_tensor(recv, send, group=group)
        return
    world = dist.get_world_size(group)
    n = send.numel()
    if send.is_cuda:
        hs = send.cpu()
        parts = [torch.empty_like(hs) for _ in range(world)]
        dist.all_gather(parts, hs, group=group)
        recv.copy_(torch.cat(parts))
    else:
        dist.all_gather([recv[r * n:(r + 1) * n] for r in range(world)], send.clone(), group=group)


class _Stitcher:
    """All-gather of the per-rank row blocks of c, with the p + bc tail rows (final-time path and boundary constraints) taken
    from the LAST rank -- the only one that holds everything they read when the iterate is sharded.  Every rank sends its
    block padded to the longest one (+ the tail slot), ONE all_gather_into_tensor, then one index kernel writes the whole c
    (index built once).  Equal blocks without tail rows are gathered in place."""

    def __init__(self, N, cb, ncon, world, rank, device, group=None):
        self.N, self.cb, self.world, self.rank, self.group = N, cb, world, rank, group
        self.tail = ncon - N * cb
        self.in_place = (N % world == 0) and self.tail == 0
        if not self.in_place and (world > 1 or _FORCE):
            blocks = [shard_steps(N, world, r) for r in range(world)]
            self.smax = max(e - b for b, e in blocks) * cb + self.tail
            self.begin, self.end = blocks[rank][0] * cb, blocks[rank][1] * cb
            idx = torch.empty(ncon, dtype=torch.long)
            for r, (b, e) in enumerate(blocks):
                idx[b * cb:e * cb] = r * self.smax + torch.arange((e - b) * cb)
            last_rows = (blocks[-1][1] - blocks[-1][0]) * cb
            idx[N * cb:] = (world - 1) * self.smax + last_rows + torch.arange(self.tail)
            self.idx = idx.to(device)
            self.send = torch.zeros(self.smax, dtype=torch.float64, device=device)
            self.recv = torch.zeros(world * self.smax, dtype=torch.float64, device=device)

    def __call__(self, c):
        if self.world == 1 and not _FORCE:
            return c
        N, cb, world, rank = self.N, self.cb, self.world, self.rank
        if self.in_place:
            S = (N // world) * cb
            body = c[:N * cb]
            _all_gather_into(body, body[rank * S:(rank + 1) * S], self.group)
            return c
        own = self.end - self.begin
        self.send[:own].copy_(c[self.begin:self.end])
        if rank == world - 1 and self.tail:
            self.send[own:own + self.tail].copy_(c[N * cb:])
        _all_gather_into(self.recv, self.send, self.group)
        torch.index_select(self.recv, 0, self.idx, out=c)
        return c


def stitch_constraints(c, N, cb, world, rank, group=None):
    """All-gather the per-rank row blocks of `c`.  `c` is the full-length constraint vector in which this rank has already
    written its step rows [begin*cb, end*cb); the tail rows [N*cb, ncon) are taken from the last rank.  Returns c.
    (One-shot form; `ShardedDOCP` keeps a `_Stitcher` so the index is built only once.)"""
    return _Stitcher(N, cb, c.numel(), world, rank, c.device, group)(c)


def reduce_objective(partial, group=None, device=None):
    """Sum of the per-shard objective partials (Lagrange partial sums; the last shard adds the Mayer term).  `partial`: a
    float, or a 1-element tensor that is reduced in place and returned (no host round trip)."""
    if torch.is_tensor(partial):
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
        return partial
    t = torch.tensor([partial], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0])


def reduce_hessian_vv(vals, vv_idx, group=None):
    """Adds the shards' partial sums of the variable x variable Hessian entries (the only entries of hess_coord! that sum
    over every time step): one all-reduce of nv (nv+1)/2 doubles, in place.  `vv_idx` from DOCP.hess_shard_info(), or the
    same positions as a long tensor on vals' device."""
    if len(vv_idx) == 0:
        return vals
    idx = vv_idx if torch.is_tensor(vv_idx) else torch.as_tensor(vv_idx, dtype=torch.long, device=vals.device)
    part = vals.index_select(0, idx)
    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    vals.index_copy_(0, idx, part)
    return vals


class ShardedDOCP:
    """One rank's view of a grid-sharded transcription.  `make_docp(steps=(begin, end))` builds the engine handle
    (ctdirect DOCP) for the rank's block of steps."""

    def __init__(self, make_docp, N, world=None, rank=None, group=None):
        self.world = dist.get_world_size(group) if world is None else world
        self.rank = dist.get_rank(group) if rank is None else rank
        self.group = group
        self.N = N
        self.steps = shard_steps(N, self.world, self.rank)
        self.docp = make_docp(steps=self.steps)
        d = self.docp
        disc = d.discretization
        self.cb = disc._state_stage_eqs_block + disc._step_pathcons_block
        self.blk = disc._step_variables_block
        self.n, self.m = d.dims.NLP_x, d.dims.NLP_u
        self.halo_w = self.n + (self.m if getattr(disc, "_final_control", False) else 0)     # trapeze: the next node's control too
        # one-point schemes (midpoint, Euler): the residual of the step BEFORE this rank's first one depends on this rank's
        # first state, and the rank owns those Jacobian entries: it needs that step's block too
        self.halo_lo = self.blk if (disc.stage == 0 and not getattr(disc, "_final_control", False)) else 0
        self._dev = None
        self._stream = None
        if d.device is not None and d.device >= 0:
            self._dev = torch.device("cuda", d.device)
            self._rebind()                  # launch on torch's current stream: ordered with the collectives issued here
        self._stitch = None
        self._halo = None
        self._yhalo = None
        self._vv = None
        self._f = None
        self._peer = None                   # data_ptr of the x buffer the peer table is ACTIVE for (enable_peer_x), None: off
        self._peer_tables = {}              # x data_ptr -> (the tensor itself, mapped pointers of the other ranks' buffers): the
                                            # tensor is kept alive so the allocator cannot hand its address to another buffer
                                            # while the other ranks still hold a mapping of it
        self._ipc_bases = []                # (device, base) of every IPC mapping this object opened
        self._bind_gen = 0                  # bumped whenever the handle's iterate mode changes: bound callables re-assert theirs

    def _rebind(self):
        """The engine's kernels and the collectives issued here must share a stream: follow torch's current stream (a cheap
        integer compare per call; a caller inside `with torch.cuda.stream(s)` or a graph capture gets both on s)."""
        if self._dev is None:
            return
        cur = torch.cuda.current_stream(self._dev).cuda_stream
        if cur != self._stream:
            self.docp.set_stream(cur)
            self._stream = cur

    # ---- iterate distribution ------------------------------------------------------------------------------------------
    def owned_variables(self):
        """[begin, end) of the entries of x this rank owns: the blocks of its own steps; the last rank also the final state
        (+ final control) -- the optimisation variables v at the tail are replicated on every rank."""
        b, e = self.steps
        end = e * self.blk
        if self.rank == self.world - 1:
            end = self.docp.dim_NLP_variables - self.docp.dims.NLP_v
        return b * self.blk, end

    def exchange_halo(self, x):
        """Sharded iterate: fills, inside this rank's full-length x, the few entries other ranks own that its rows and its
        Jacobian columns read -- the next rank's first node (X, + U for trapeze); for the one-point schemes (midpoint, Euler)
        the previous rank's last step block, whose residual depends on this rank's first state; X_1 and X_{N+1} (boundary rows,
        Mayer cost) -- with ONE all-gather of (halo_w + low + n) doubles per rank.  v is replicated by the solver and not
        touched.  Three small kernels around the collective: pack (index_select), unpack (index_select + index_copy_)."""
        if self.world == 1 and not _FORCE:
            return x
        if self._halo is None:
            w, n, blk, N, lo = self.halo_w, self.n, self.blk, self.N, self.halo_lo
            L = w + lo + n
            b, e = self.steps
            r, G = self.rank, self.world
            ar = torch.arange
            pack = torch.cat([b * blk + ar(w), (e - 1) * blk + ar(lo), N * blk + ar(n)])      # my first node | last block | final state
            dst, src = [], []
            if r + 1 < G:
                dst += [e * blk + ar(w), N * blk + ar(n)]
                src += [(r + 1) * L + ar(w), (G - 1) * L + w + lo + ar(n)]
            if r > 0:
                if lo:
                    dst.append((b - 1) * blk + ar(lo))
                    src.append((r - 1) * L + w + ar(lo))
                dst.append(ar(n))
                src.append(ar(n))
            empty = torch.zeros(0, dtype=torch.long)
            self._halo = (pack.to(x.device), (torch.cat(dst) if dst else empty).to(x.device), (torch.cat(src) if src else empty).to(x.device),
                          torch.zeros(L, dtype=torch.float64, device=x.device), torch.zeros(G * L, dtype=torch.float64, device=x.device))
        pack, dst, src, send, recv = self._halo
        torch.index_select(x, 0, pack, out=send)
        _all_gather_into(recv, send, self.group)
        if dst.numel():
            x.index_copy_(0, dst, recv.index_select(0, src))
        return x

    def owned_constraints(self):
        """[begin, end) of the rows of c -- and of the multipliers y -- this rank owns: the rows of its own steps; the last rank also
        the tail (final path rows + boundary rows), as in `stitch_constraints`."""
        b, e = self.steps
        return b * self.cb, (self.docp.dim_NLP_constraints if self.rank == self.world - 1 else e * self.cb)

    def exchange_multipliers(self, y):
        """Sharded multipliers for `hess_coord`: fills, inside this rank's full-length y, the rows other ranks own that its Hessian
        entries read -- the PREVIOUS rank's last step (the second-order terms of that step's defect in this rank's first node:
        trapeze, midpoint, implicit Euler; the Gauss-Legendre schemes read none) and the tail rows (final path + boundary
        multipliers, owned by the last rank) -- with ONE all-gather of (cb + tail) doubles per rank, instead of a replicated y
        (an all-gather of all ncon rows).  Entries: /root/reference/src/DOCP_functions.jl:80-140 rows, multiplied into the
        Lagrangian by hess_coord!(nlp, x, y, vals; obj_weight)."""
        if self.world == 1 and not _FORCE:
            return y
        if self._yhalo is None:
            cb, N = self.cb, self.N
            tail = self.docp.dim_NLP_constraints - N * cb
            b, e = self.steps
            r, G = self.rank, self.world
            ar = torch.arange
            Ly = cb + tail
            pack = torch.cat([(e - 1) * cb + ar(cb), N * cb + ar(tail)])            # my last step | the tail (meaningful on the last rank)
            dst, src = [], []
            if r > 0:
                dst.append((b - 1) * cb + ar(cb))
                src.append((r - 1) * Ly + ar(cb))
            if r + 1 < G and tail:
                dst.append(N * cb + ar(tail))
                src.append((G - 1) * Ly + cb + ar(tail))
            empty = torch.zeros(0, dtype=torch.long)
            self._yhalo = (pack.to(y.device), (torch.cat(dst) if dst else empty).to(y.device), (torch.cat(src) if src else empty).to(y.device),
                           torch.zeros(Ly, dtype=torch.float64, device=y.device), torch.zeros(G * Ly, dtype=torch.float64, device=y.device))
        pack, dst, src, send, recv = self._yhalo
        torch.index_select(y, 0, pack, out=send)
        _all_gather_into(recv, send, self.group)
        if dst.numel():
            y.index_copy_(0, dst, recv.index_select(0, src))
        return y

    def enable_peer_x(self, x):
        """Sharded iterate read IN PLACE (`ctd_set_x_shards`): every rank exports its full-length x buffer once (IPC handle,
        all-gathered as Python objects -- set-up, not the step), maps the other ranks' buffers, and from then on this rank's
        constraint / Jacobian kernels load the few entries its neighbours own (next rank's first node, previous rank's last
        block, X_1, X_{N+1}) straight from their HBM over xGMI.  The evaluation of a step then contains NO collective, no
        pack / unpack kernel and no copy.  `x` must stay the iterate buffer (same storage) for the life of this object; the
        solver's update of x on every rank has to be complete before the evaluations that follow it are enqueued (its step
        acceptance is a collective anyway).  The shard table is state of the HANDLE: while it is active, `cons_jac`, `obj` and
        `hess_coord` and `grad` read the neighbours' entries in place too and must be given this same x (anything else raises);
        `docp.grad` (the C ABI's ctd_grad*) stays the whole objective's gradient and needs an all-gathered x.  `disable_peer_x`
        (or binding another x_mode) switches back."""
        from . import _lib
        import ctypes as C
        if self.world == 1:
            return x
        begins = [shard_steps(self.N, self.world, r)[0] for r in range(self.world)] + [self.N]
        if self._peer == x.data_ptr():
            return x
        if x.data_ptr() in self._peer_tables:            # mapped before (the caller alternates between modes)
            self.docp.set_x_shards(begins, self._peer_tables[x.data_ptr()][1], self.rank)
            self._peer = x.data_ptr()
            self._bind_gen += 1
            return x
        L = _lib.lib()
        dev = x.device.index
        hbuf = (C.c_ubyte * 64)()
        off = C.c_int64()
        # Every rank goes through the same collectives whatever fails locally (a rank that raised early would leave the
        # others waiting): errors are collected, agreed on with one all-reduce, and raised on EVERY rank together.
        err = None
        st = L.ctd_ipc_export(dev, C.c_void_p(x.data_ptr()), hbuf, C.byref(off))
        if st:
            err = "ctd_ipc_export: " + L.ctd_last_error(None).decode()
        mine = (bytes(hbuf), int(off.value), os.getpid(), int(x.data_ptr()), err is None)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        ptrs, opened = [], []
        if all(e[4] for e in everyone):
            for r, (hb, o, pid, raw, _) in enumerate(everyone):
                if r == self.rank:
                    ptrs.append(0)
                elif pid == os.getpid():      # same process (several ranks of a test harness): the pointer is valid as it is
                    ptrs.append(raw)
                else:
                    base = C.c_void_p()
                    st = L.ctd_ipc_open(dev, (C.c_ubyte * 64).from_buffer_copy(hb), C.byref(base))
                    if st:
                        err = f"ctd_ipc_open (rank {r}'s buffer): " + L.ctd_last_error(None).decode()
                        break
                    opened.append((dev, base.value))
                    ptrs.append(base.value + o)
                    # one read through the mapping by a plain device copy: an unreachable mapping is an error here, not a fault
                    # inside the evaluation kernel
                    if L.ctd_ipc_probe(dev, C.c_void_p(base.value + o), 8):
                        err = f"ctd_ipc_probe (rank {r}'s buffer): " + L.ctd_last_error(None).decode()
                        break
        elif err is None:
            err = "another rank could not export its buffer"
        flat = str(dist.get_backend(self.group)).lower() in ("nccl", "rccl")
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=x.device if flat else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)     # also: every rank has mapped the others before anyone evaluates
        if ok.item() != 1.0:
            for d_, b_ in opened:
                L.ctd_ipc_close(d_, b_)
            raise RuntimeError("enable_peer_x: " + (err or "another rank could not map the buffers"))
        self._ipc_bases += opened
        self.docp.set_x_shards(begins, ptrs, self.rank)
        self._peer_tables[x.data_ptr()] = (x, ptrs)
        self._peer = x.data_ptr()
        self._bind_gen += 1
        return x

    def disable_peer_x(self):
        """Back to "the x passed to a call holds everything the rank reads" (the mappings stay open for a later enable_peer_x)."""
        if self._peer is not None:
            self.docp.set_x_shards(None, None, 0)
            self._peer = None
            self._bind_gen += 1

    def _check_peer(self, x, what):
        """While the shard table is active the kernels take a rank's own entries from the x of the call and its neighbours'
        entries from the buffers registered with enable_peer_x: any other x would silently mix two iterates."""
        if self._peer is not None and torch.is_tensor(x) and x.data_ptr() != self._peer:
            raise ValueError(f"ShardedDOCP.{what}: the sharded iterate is read in place from the buffers registered with "
                             "enable_peer_x; pass that same x, or call disable_peer_x() first")

    def broadcast_iterate(self, x, src=0):
        """Replicated iterate: the rank that holds the new x sends all of it (nvar doubles) to every other rank."""
        if self.world > 1 or _FORCE:
            dist.broadcast(x, src=src, group=self.group)
        return x

    # ---- callbacks -------------------------------------------------------------------------------------------------------
    def _stitcher(self, c):
        if self._stitch is None:
            self._stitch = _Stitcher(self.N, self.cb, c.numel(), self.world, self.rank, c.device, self.group)
        return self._stitch

    def cons_jac(self, x, c, vals, stitch=True):
        """Evaluate this rank's rows into the full-length c / vals buffers; `stitch`: all-gather the row blocks of c so that
        every rank holds the whole residual (the Jacobian values always stay sharded)."""
        self._rebind()
        self._check_peer(x, "cons_jac")
        self.docp.cons_jac(x, c, vals, sync=False)
        if stitch:
            self._stitcher(c)(c)
        return c, vals

    def bind_cons_jac(self, x, c, vals, stitch=True, x_mode=None):
        """Zero-argument callable for a solver loop, pointers pre-bound: [distribute the iterate: x_mode "peer" (sharded x,
        neighbours' entries read in place by the kernel: nothing per step) / "halo" (sharded x, exchange_halo: one all-gather) /
        "broadcast" (replicated x from rank 0) / None] + enqueue this rank's evaluation [+ the all-gather of c
        when `stitch`]."""
        self._rebind()
        launch = self.docp.bind_cons_jac(x, c, vals, sync=False)
        if self.world == 1 and not _FORCE:
            return launch
        def assert_mode():                   # (the shard table is per-handle state: the callable bound LAST owns it ...)
            if x_mode == "peer":             # neighbours' entries are read in place by the kernel: nothing precedes the launch
                self.enable_peer_x(x)        # (first time: collective set-up; afterwards a table lookup)
            else:
                self.disable_peer_x()
            return self._bind_gen
        gen = [assert_mode()]
        pre = {None: None, "peer": None, "halo": self.exchange_halo, "broadcast": self.broadcast_iterate}[x_mode]
        post = self._stitcher(c) if stitch else None

        def call():
            if gen[0] != self._bind_gen:     # (... and an older callable that runs again puts its own mode back first)
                gen[0] = assert_mode()
            if pre is not None:
                pre(x)
            launch()
            if post is not None:
                post(c)
        return call

    def hess_coord(self, x, y, obj_weight, vals):
        """This rank's entries of hess_coord!(nlp, x, y, vals; obj_weight) into the full-length `vals` (they stay sharded like
        the Jacobian values), plus the all-reduced variable x variable entries on every rank.  No host synchronisation: the
        kernel and the all-reduce are ordered on torch's current stream.  `y` is full-length; only this rank's own rows, the
        previous rank's last step and the tail rows are read (`exchange_multipliers` fills the latter two of a sharded y)."""
        self._rebind()
        self._check_peer(x, "hess_coord")
        self.docp.hess_coord(x, y, obj_weight, vals, sync=False)
        if self.world > 1 or _FORCE:
            if self._vv is None:
                self._vv = torch.as_tensor(self.docp.hess_shard_info()[2], dtype=torch.long, device=vals.device)
            reduce_hessian_vv(vals, self._vv, self.group)
        return vals

    def grad(self, x, g):
        """grad!(nlp, x, g) of the sharded transcription: this rank's own entries of g (its step blocks; the last rank also the
        final state) into the full-length g, from the sharded iterate read in place (or with copied halos) -- no all-gathered x --
        plus ONE all-reduce of the nv entries of d/dv, which sum over every step.  Entries other ranks own are left untouched."""
        self._rebind()
        self._check_peer(x, "grad")
        self.docp.grad_shard(x, g)
        nv = self.docp.dims.NLP_v
        if nv and (self.world > 1 or _FORCE):
            tail = g[g.numel() - nv:]
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
        return g

    def obj(self, x, as_tensor=False):
        """Objective of the whole transcription: the shards' partial sums added with one all-reduce of one double that never
        leaves the device; `as_tensor=False` reads the result back at the end (the only host round trip)."""
        if not torch.is_tensor(x):
            return reduce_objective(self.docp.obj(x), self.group)
        if self._f is None:
            self._f = torch.zeros(1, dtype=torch.float64, device=x.device)
        self._rebind()
        self._check_peer(x, "obj")
        self.docp.obj_async(x, self._f)
        if self.world > 1 or _FORCE:
            dist.all_reduce(self._f, op=dist.ReduceOp.SUM, group=self.group)
        return self._f if as_tensor else float(self._f.item())

    def close(self):
        if self._ipc_bases:
            from . import _lib
            torch.cuda.synchronize(self._dev)
            for dev, b in self._ipc_bases:
                _lib.lib().ctd_ipc_close(dev, b)
            self._ipc_bases = []
        self._peer = None
        self._peer_tables = {}
        self.docp.close()
