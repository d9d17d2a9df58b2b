"""Multi-GPU data parallelism for the object branch: one process per GPU, rays sharded across ranks, RCCL over
xGMI via torch.distributed (backend "nccl" == RCCL on ROCm).

The reference has no distributed path on the live loop; this module introduces the one exchange
step the path needs, shaped for MI355X's point-to-point xGMI fabric (a ring collective is bound by ONE ~50-64 GB/s
link at W=2 and by a few links at W=8, so bytes on the wire are what matters).  Two modes:

mode "samples" (default) - exchange the k0 gradient at SAMPLE granularity
  * a rank's rays reach the dense 196 MB gradient grid through ~55 k samples only, so what travels is the INPUT of the
    scatter: 64 B per sample (12 feature gradients + position), one all-gather of [rows, 16] floats per rank instead of a
    196 MB reduce-scatter plus a 196 MB all-gather.  `rows` is EXACT for every step: right after the sampler each rank
    publishes (sample count, masked-pixel count) in a 8-byte all-gather that completes under the forward pass; the host
    reads the counts from pinned memory when it enqueues the exchange (a wait on an event the GPU passed ~0.5 ms of queued
    work earlier - the launch queue stays full) and sends the largest count rounded up to 1024 rows (3.6 MB per rank at
    the bench workload instead of the 12 MB worst case).  No sample is ever dropped, no calibration window exists; the
    exchange overlaps the geometry / warp backward,
  * the same 8 bytes normalise the losses over the UNION batch (lib/losses.py divides the masked MSE by the batch's
    masked-pixel count and the sample priors by the batch's sample count), so W ranks x N rays reproduce one W*N-ray step,
  * every rank replays the scatter for all shards (15 us each) and runs the full fused TV+Adam pass (replicated),
  * no parameter all-gather.  Float atomics make the replicas differ in the last bit, so every `resync_every` steps
    rank 0 broadcasts grid + moments (amortised to ~20 us / step).

mode "zero1" (PP_DIST_MODE=zero1) - dense exchange with a sharded optimiser
  * dense k0 gradient  : reduce-scatter along X (each rank receives the sum of its own x-slab only),
  * optimiser          : ZeRO-1 - each rank runs the fused TV+Adam kernel on its slab (1/W of the dense traffic),
  * parameters         : all-gather of the updated slabs (bit-identical replicas by construction).

Both: everything small goes through ONE all-reduce bucket (MLPs + sdf alpha/beta + 6-DoF pose grads, ~370 KB); the
TV term is rank invariant (a function of the replicated parameters) and is added inside the optimiser kernel, never
reduced; gradients are averaged (sum * 1/W), i.e. the global batch is W * N_rand rays.
"""
import os

import torch
import torch.distributed as dist


def slab_bounds(X, world, rank):
    """Equal x-slabs [begin, end); requires X % world == 0 for the sharded path."""
    per = X // world
    return rank * per, (rank + 1) * per


def default_mode(world):
    """The exchange mode a world size runs by default (DESIGN.md 7, table "predicted step time"): from the measured kernel
    times of the 160^3 / 1024-ray step and a 50-60 GB/s usable xGMI link, "samples" (3.6 MB per rank on the wire, replicated
    248-295 us grid pass, W x 17 us replayed scatters) beats "zero1" (2 x 196 MB / W per link, grid pass / W) at every W <= 8:
    1.13 / 1.18 / 1.26 ms against 4.0 / 2.1-5.7 / 1.2-6.5 ms at W = 2 / 4 / 8 - zero1 only comes close at W = 8 and only if RCCL
    really drives all seven links at once.  Beyond one node (W > 8) the replay grows linearly and the dense pass is 1/W: zero1.
    PP_DIST_MODE overrides; the driver's SCALE run is the measurement that settles it."""
    return 'samples' if world <= 8 else 'zero1'


class DistContext:
    def __init__(self, group=None, mode=None, resync_every=256, row_quantum=1024):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self._bucket = None
        self.mode = mode or os.environ.get('PP_DIST_MODE') or default_mode(self.world)
        assert self.mode in ('samples', 'zero1')
        self.resync_every = resync_every
        self._gathered = None
        self._steps = 0
        # mode "samples": rows of the packed buffer that travel = the step's largest per-rank sample count, rounded up
        self.row_quantum = row_quantum
        self.rows = None                   # rows exchanged by the latest step (diagnostics / tests)
        self.batch_norm = None             # device float[2]: union-batch normalisers of the losses (see start_batch_stats)
        self._stats = None

    # ---- generic tensor-level collectives (also exercised on CPU with gloo) ---------------------------------
    def shardable(self, X):
        return X % self.world == 0

    def reduce_scatter_grid(self, grid_grad):
        """grid_grad [X, ...] contiguous. On return the caller's slab holds SUM over ranks; the rest is zeroed."""
        X = grid_grad.shape[0]
        xb, xe = slab_bounds(X, self.world, self.rank)
        if self.backend == 'gloo':          # gloo has no reduce_scatter: all_reduce + slice (CPU rehearsal only)
            dist.all_reduce(grid_grad, group=self.group)
        else:
            dist.reduce_scatter_tensor(grid_grad[xb:xe], grid_grad, op=dist.ReduceOp.SUM, group=self.group)
        if xb > 0:
            grid_grad[:xb].zero_()
        if xe < X:
            grid_grad[xe:].zero_()
        return xb, xe

    def all_gather_grid(self, grid):
        """Every rank contributes its own x-slab of `grid` (in place)."""
        X = grid.shape[0]
        xb, xe = slab_bounds(X, self.world, self.rank)
        if self.backend == 'gloo':
            parts = [torch.empty_like(grid[xb:xe]) for _ in range(self.world)]
            dist.all_gather(parts, grid[xb:xe].contiguous(), group=self.group)
            for r, p in enumerate(parts):
                b, e = slab_bounds(X, self.world, r)
                grid[b:e].copy_(p)
        else:
            dist.all_gather_into_tensor(grid, grid[xb:xe], group=self.group)

    def all_reduce_small(self, tensors, async_op=False):
        """One bucketed all-reduce (SUM) over a list of small tensors.  async_op=True: the reduced values are copied back by
        wait_small() - the engine calls it after the dense grid pass, so the (latency-bound) collective hides behind it."""
        n = sum(t.numel() for t in tensors)
        if self._bucket is None or self._bucket.numel() != n or self._bucket.device != tensors[0].device:
            self._bucket = torch.empty(n, dtype=tensors[0].dtype, device=tensors[0].device)
        o = 0
        for t in tensors:
            self._bucket[o:o + t.numel()].copy_(t.reshape(-1))
            o += t.numel()
        work = dist.all_reduce(self._bucket, group=self.group, async_op=async_op)
        self._small = (work, tensors)
        if not async_op:
            self.wait_small()

    def wait_small(self):
        pending = getattr(self, '_small', None)
        if pending is None:
            return
        work, tensors = pending
        self._small = None
        if work is not None:
            work.wait()
        o = 0
        for t in tensors:
            t.copy_(self._bucket[o:o + t.numel()].view_as(t))
            o += t.numel()

    def all_reduce_tensor(self, t):
        """In-place SUM all-reduce of one contiguous tensor (the scene networks' packed gradient blocks)."""
        dist.all_reduce(t, group=self.group)

    def all_gather_rows(self, local, out=None, async_op=False):
        """local [rows, ld] -> out [W, rows, ld] (every rank's buffer, rank order).  Returns (out, work)."""
        if out is None:
            out = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        if self.backend == 'gloo':
            parts = [out[r] for r in range(self.world)]
            work = dist.all_gather(parts, local.contiguous(), group=self.group, async_op=async_op)
        else:
            work = dist.all_gather_into_tensor(out, local, group=self.group, async_op=async_op)
        return out, work

    def broadcast_state(self, tensors, src=0):
        for t in tensors:
            dist.broadcast(t, src=src, group=self.group)

    # ---- hooks used by engine.TrainEngine (overlap of the big collectives with compute) ---------------------------
    @property
    def local_scatter(self):
        """False: the engine must not scatter its own samples into k0_grad (mode "samples" replays all shards later)."""
        return self.mode != 'samples'

    def start_grid_reduce(self, eng):
        """Called right after the colour-feature backward.  "samples": pack this rank's scatter input and start the
        async all-gather; "zero1": start the dense reduce-scatter.  Either overlaps the geometry / warp-MLP backward,
        which does not touch the grid gradient."""
        if self.mode == 'samples':
            from . import ops
            ws = eng.ws
            rows = self.rows = self.exchange_rows(self.step_counts().max(), ws.cap)
            if getattr(ws, 'k0_packed', None) is None:
                ws.k0_packed = torch.zeros(ws.cap, 16, dtype=torch.float32, device=ws.pts.device)
            if self._gathered is None or self._gathered.numel() != self.world * ws.cap * 16:
                self._gathered = torch.zeros(self.world * ws.cap * 16, dtype=torch.float32, device=ws.pts.device)
            ops.k0_pack_samples(ws.pts, ws.g_feat, ws.count, rows, eng.cfg.k0_dim, ws.k0_packed)
            self._gathered_view = self._gathered[:self.world * rows * 16].view(self.world, rows, 16)
            _, self._grid_work = self.all_gather_rows(ws.k0_packed[:rows], self._gathered_view, async_op=True)
            return
        X = eng.k0_grad.shape[0]
        if self.shardable(X) and self.backend != 'gloo':
            xb, xe = slab_bounds(X, self.world, self.rank)
            self._grid_work = dist.reduce_scatter_tensor(eng.k0_grad[xb:xe], eng.k0_grad, op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True)
        else:
            self._grid_work = dist.all_reduce(eng.k0_grad, group=self.group, async_op=True)

    def reduce_gradients(self, eng):
        X = eng.k0_grad.shape[0]
        if self.mode == 'samples':
            from . import ops
            self._grid_work.wait()
            self._grid_work = None
            rows = self._gathered_view.shape[1]
            if getattr(eng, 'deterministic_scatter', False):      # replicas stay bit-identical: every rank adds in (rank, sample) order
                ops.k0_scatter_packed_sorted(eng.cfg.pp, self._gathered_view, self.world, rows, eng.k0_grad,
                                             eng.scatter_work(self.world * rows), eng.k0_touched[eng.touch_par])
            else:
                ops.k0_scatter_packed(eng.cfg.pp, self._gathered_view, self.world, rows, eng.k0_grad, eng.k0_touched[eng.touch_par])
            eng.x_slab = (0, X)
            self.all_reduce_small([eng.flat.grad, eng.se3_grad], async_op=True)     # finished by wait_small() after the grid pass
            eng.grad_scale = 1.0 / self.world
            return
        work = getattr(self, '_grid_work', None)
        if work is None:
            self.start_grid_reduce(eng)
            work = self._grid_work
        work.wait()
        self._grid_work = None
        if self.shardable(X):
            xb, xe = slab_bounds(X, self.world, self.rank)
            if xb > 0:
                eng.k0_grad[:xb].zero_()
            if xe < X:
                eng.k0_grad[xe:].zero_()
            eng.x_slab = (xb, xe)
        else:
            eng.x_slab = (0, X)
        self.all_reduce_small([eng.flat.grad, eng.se3_grad], async_op=True)
        eng.grad_scale = 1.0 / self.world

    def gather_parameters(self, eng):
        """Async all-gather of the updated slabs; the next step waits for it only right before its first k0 lookup."""
        X = eng.k0_grad.shape[0]
        self._param_work = None
        if self.mode == 'samples':
            self._steps += 1
            resync = self.resync_every and self._steps % self.resync_every == 0
            if resync:
                self.broadcast_state([eng.k0_cl, eng.k0_m, eng.k0_v])    # replicas differ in the last bit (float atomics)
            return
        if self.shardable(X):
            if self.backend == 'gloo':
                self.all_gather_grid(eng.k0_cl)
            else:
                xb, xe = slab_bounds(X, self.world, self.rank)
                self._param_work = dist.all_gather_into_tensor(eng.k0_cl, eng.k0_cl[xb:xe], group=self.group,
                                                               async_op=True)

    def exchange_rows(self, max_count, cap):
        """Rows to exchange for the step's largest per-rank count: rounded up to the row quantum, at least one quantum
        (row 0 carries the count), never above the capacity (the sampler clamps counts to it)."""
        q = self.row_quantum
        return int(min(cap, max(q, -(-int(max_count) // q) * q)))

    # ---- per-step batch statistics: sizes the sample exchange and normalises the losses over the union batch -----------
    def start_batch_stats(self, count, mask_px):
        """Called right after the sampler.  Publishes this rank's (masked-pixel count, sample count) to every rank
        (all-gather of 2 floats, asynchronous: it completes under the forward pass), leaves `batch_norm` = column sums / W on
        the device for the loss kernels (pp_loss_rays / pp_geometry_bwd_priors) and the per-rank counts in pinned host
        memory for step_counts().  Counts are exact in fp32 (< 2^24 samples per rank)."""
        dev = count.device
        if self._stats is None or self._stats['local'].device != dev:
            f = dict(dtype=torch.float32, device=dev)
            st = self._stats = dict(local=torch.zeros(2, **f), all=torch.zeros(self.world, 2, **f), norm=torch.zeros(2, **f),
                                    host=torch.zeros(self.world, 2, dtype=torch.float32))
            if dev.type == 'cuda':
                st['host'] = st['host'].pin_memory()
                st['side'], st['ready'], st['landed'] = torch.cuda.Stream(dev), torch.cuda.Event(), torch.cuda.Event()
        st = self._stats
        torch.sum(mask_px.reshape(-1), dim=0, keepdim=True, out=st['local'][0:1])
        st['local'][1:2].copy_(count)
        _, work = self.all_gather_rows(st['local'], st['all'], async_op=(dev.type == 'cuda'))
        if dev.type == 'cuda':
            # everything downstream of the tiny collective runs on a side stream, so the compute stream never waits for it
            # before it has to (the loss kernels, a whole forward pass later)
            with torch.cuda.stream(st['side']):
                work.wait()
                torch.sum(st['all'], dim=0, out=st['norm'])
                st['norm'].mul_(1.0 / self.world)
                st['ready'].record()
                st['host'].copy_(st['all'], non_blocking=True)
                st['landed'].record()
        else:
            torch.sum(st['all'], dim=0, out=st['norm'])
            st['norm'].mul_(1.0 / self.world)
            st['host'].copy_(st['all'])
        self.batch_norm = st['norm']

    def wait_batch_stats(self):
        """Compute stream waits (device side, no host involvement) until `batch_norm` is valid; -> batch_norm."""
        st = self._stats
        if st is not None and 'ready' in st:
            torch.cuda.current_stream().wait_event(st['ready'])
        return self.batch_norm

    def step_counts(self):
        """Per-rank sample counts of the current step (host tensor [W]).  The only host-side wait of the mode: on an event
        recorded right after the sampler, i.e. the GPU is by then at most one forward pass behind the host."""
        st = self._stats
        if 'landed' in st:
            st['landed'].synchronize()
        return st['host'][:, 1]

    def wait_parameters(self, eng):
        work = getattr(self, '_param_work', None)
        if work is not None:
            work.wait()
            self._param_work = None
