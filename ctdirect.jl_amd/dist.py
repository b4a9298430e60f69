"""Time-step sharding of one transcription across the GPUs of a node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no parallelism of any kind; what is partitioned here is its main loop
`for i in 1:docp.time.steps` (src/DOCP_functions.jl:92-98): step i reads X_i, U_i, K_i, X_{i+1} and v, and writes only
its own rows c[(i-1)(eqs+p)+1 : i(eqs+p)] and the Jacobian entries of those rows.

  * x is replicated (it is the solver's iterate; 1e6 doubles = 8 MB), so no halo exchange is needed;
  * rank r evaluates the contiguous block of steps [r N/G, (r+1) N/G); the last rank also owns the final-time path rows
    and the boundary rows, the first rank the irregular first-step columns;
  * each rank writes its rows straight into a full-length c buffer at their global position, and every rank computes the
    p+bc tail rows (final-time path and boundary constraints) itself;
  * outputs stay ROW-SHARDED: rank r holds its rows of c and, for the Jacobian values, one contiguous range of the global
    CSC value array (its step columns) plus its slice of every V column -- what a distributed KKT consumer wants;
    `DOCP.shard` gives the ranges.  The evaluation itself needs no collective;
  * a consumer that wants the residual vector whole on every rank asks for it (`stitch=True`): ONE in-place all-gather of
    the row blocks per evaluation.
"""
import torch
import torch.distributed as dist


def shard_steps(N, world, rank):
    """Contiguous block of ceil/floor(N / world) steps for `rank` (equal blocks when world divides N)."""
    base, rem = divmod(N, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def stitch_constraints(c, N, cb, world, rank, group=None):
    """All-gather the per-rank row blocks of `c` in place.  `c` is the full-length constraint vector in which this
    rank has already written its step rows [begin*cb, end*cb) and the tail rows [N*cb, ncon) (every rank computes
    those).  Returns c.

    Equal blocks (world | N): ONE all_gather_into_tensor over c[:N*cb] with the rank's own block as the send buffer
    (in place).  Ragged blocks fall back to one broadcast per rank."""
    if world == 1:
        return c
    if N % world == 0:
        S = (N // world) * cb
        body = c[:N * cb]
        try:
            dist.all_gather_into_tensor(body, body[rank * S:(rank + 1) * S], group=group)
        except (RuntimeError, NotImplementedError):
            dist.all_gather([body[r * S:(r + 1) * S] for r in range(world)], body[rank * S:(rank + 1) * S].clone(), group=group)
    else:
        for r in range(world):
            b, e = shard_steps(N, world, r)
            dist.broadcast(c[b * cb:e * cb], src=r, group=group)
    return c


def reduce_objective(partial, group=None, device=None):
    """Sum of the per-shard objective partials (Lagrange partial sums; the last shard adds the Mayer term)."""
    t = torch.tensor([partial], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0])


def reduce_hessian_vv(vals, vv_idx, group=None):
    """Adds the shards' partial sums of the variable x variable Hessian entries (the only entries of hess_coord! that sum
    over every time step): one all-reduce of nv (nv+1)/2 doubles, in place.  `vv_idx` from DOCP.hess_shard_info()."""
    if len(vv_idx) == 0:
        return vals
    idx = torch.as_tensor(vv_idx, dtype=torch.long, device=vals.device)
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
        disc = self.docp.discretization
        self.cb = disc._state_stage_eqs_block + disc._step_pathcons_block

    def cons_jac(self, x, c, vals, stitch=True):
        """Evaluate this rank's rows into the full-length c / vals buffers; `stitch`: all-gather the row blocks of c so that
        every rank holds the whole residual (the Jacobian values always stay sharded)."""
        self.docp.cons_jac(x, c, vals, sync=False)
        if stitch:
            stitch_constraints(c, self.N, self.cb, self.world, self.rank, self.group)
        return c, vals

    def bind_cons_jac(self, x, c, vals, stitch=True):
        """Zero-argument callable: enqueue this rank's evaluation (+ the all-gather of c when `stitch`), pointers pre-bound."""
        launch = self.docp.bind_cons_jac(x, c, vals, sync=False)
        if self.world == 1 or not stitch:
            return launch
        N, cb, world, rank, group = self.N, self.cb, self.world, self.rank, self.group

        def call():
            launch()
            stitch_constraints(c, N, cb, world, rank, group)
        return call

    def hess_coord(self, x, y, obj_weight, vals):
        """This rank's entries of hess_coord!(nlp, x, y, vals; obj_weight) into the full-length `vals` (they stay sharded like
        the Jacobian values), plus the all-reduced variable x variable entries on every rank."""
        self.docp.hess_coord(x, y, obj_weight, vals, sync=False)
        if self.world > 1:
            self.docp.sync()
            reduce_hessian_vv(vals, self.docp.hess_shard_info()[2], self.group)
        return vals

    def obj(self, x):
        return reduce_objective(self.docp.obj(x), self.group, device=x.device if hasattr(x, "device") else None)
