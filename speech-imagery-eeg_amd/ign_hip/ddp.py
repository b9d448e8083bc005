"""Batch-sharded data parallelism for the IGN training step: one process per GPU, one flat fp32 gradient bucket,
one RCCL all-reduce per optimizer step.

The reference has only ``nn.DataParallel`` (IGN/exp/experiment_classification.py:279-281).  Here every rank owns
a full replica and a disjoint slice of each global batch; after backward the gradients -- laid out contiguously
in ONE buffer whose slices are the parameters' ``.grad`` views, so no flatten/unflatten copies exist -- are summed
over ranks with a single ``all_reduce`` (4.27 MB for IGN-default: a latency-bound message on xGMI, one launch
instead of one per tensor) and divided by the world size.  BatchNorm statistics stay per-rank, like DataParallel.
Device-agnostic on purpose: the gloo/CPU tests in tests/test_ddp_cpu.py run the same code.
"""
import torch
import torch.distributed as dist


class FlatParamBucket:
    ALIGN = 64          # floats

    def __init__(self, module, world_size=None, process_group=None, force_collective=False):
        self.group = process_group
        # run the all-reduce even at world size 1 (a one-rank RCCL communicator: exercises initialisation, stream ordering with
        # the library's kernels and the timing fields on a one-GPU box; the result is unchanged -- sum over one rank, / 1)
        self.force_collective = bool(force_collective)
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # every parameter starts on a 256-byte boundary of the flat buffer: the HIP kernels read parameters and write
        # gradients with 16-byte vector accesses (the zero padding in between is inert for the all-reduce and for Adam)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat_grad = torch.zeros(n, device=dev, dtype=dt)
        self.views = []
        for p, off in zip(self.params, self.offsets):
            if p.device != dev or p.dtype != dt:
                raise ValueError("all parameters must share one device and dtype")
            self.views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            p.grad = None if self.flat_grad.is_cuda else self.views[-1]
        # slots that may hold something other than zeros (`gather`): the buffer starts zero-filled, and a parameter that gets no
        # gradient step after step (three of the IGN model's) needs no fill kernel per step to keep its slot at zero
        self._dirty = [False] * len(self.params)
        self.module = module

    @property
    def nbytes(self):
        return self.flat_grad.numel() * self.flat_grad.element_size()

    def broadcast_state(self, src=0):
        """Make every replica identical to rank `src` (parameters and buffers) before the first step."""
        if self.world == 1:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)

    def gather(self):
        """Bring the gradients autograd produced into the flat buffer.

        On the GPU ``zero_grad`` leaves ``p.grad = None``, so autograd hands every parameter a fresh gradient tensor (no
        accumulate kernel); one ``ign_gather_flat`` launch copies them all into their slots and ``p.grad`` is pointed back
        at the slot, so clipping, the all-reduce and the optimizer see one buffer.  Parameters whose ``.grad`` already is
        the slot view (CPU path, or a second call) are left alone; parameters without a gradient get zeros -- by a fill only when
        their slot was written since it was last zero (the buffer starts zero-filled; scaling by clipping or averaging keeps
        zeros zero)."""
        todo = []
        for i, (p, off, view) in enumerate(zip(self.params, self.offsets, self.views)):
            g = p.grad
            if g is view:
                self._dirty[i] = True                    # written in place by autograd (CPU path) or by the caller
                continue
            if g is None:
                if self._dirty[i]:
                    view.zero_()
                    self._dirty[i] = False
            else:
                todo.append((g.contiguous(), off, view))
                self._dirty[i] = True
            p.grad = view
        if not todo:
            return
        if not self.flat_grad.is_cuda:
            for g, _, view in todo:
                view.copy_(g)
            return
        import ctypes
        from . import _lib
        n = len(todo)
        src = (ctypes.c_void_p * n)(*[g.data_ptr() for g, _, _ in todo])
        off = (ctypes.c_longlong * n)(*[o for _, o, _ in todo])
        cnt = (ctypes.c_longlong * n)(*[g.numel() for g, _, _ in todo])
        _lib.check(_lib.lib().ign_gather_flat(src, off, cnt, n, ctypes.c_void_p(self.flat_grad.data_ptr()),
                                              _lib.stream()), "ign_gather_flat")

    def allreduce(self):
        """Average the gradients over ranks: one collective on the flat bucket."""
        self.gather()
        if self.world == 1 and not self.force_collective:
            return
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        if self.world > 1:
            self.flat_grad.div_(self.world)

    def zero_grad(self):
        """GPU: drop the gradients (``gather`` rebuilds the flat buffer at the next step, no zero-fill and no
        accumulate kernels).  CPU: zero in place, the gradients stay views of the flat buffer."""
        if self.flat_grad.is_cuda:
            for p in self.params:
                p.grad = None
        else:
            self.flat_grad.zero_()


def shard_indices(n, rank, world, epoch=0, seed=0, shuffle=True):
    """Global permutation from a shared seed; rank r takes the r-th contiguous slice of every global batch's worth.
    Trailing samples that do not fill all ranks evenly are dropped (equal shard sizes keep the CE means exact)."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        perm = torch.randperm(n, generator=g)
    else:
        perm = torch.arange(n)
    per = n // world
    return perm[rank * per:(rank + 1) * per]


class FlatAdam(torch.optim.Optimizer):
    """Adam over ONE flat parameter buffer: a single `ign_adam_step` launch per optimizer step instead of torch's
    per-tensor (or multi-tensor) update -- torch.optim.Adam semantics (betas (0.9, 0.999), eps 1e-8, no weight decay),
    as constructed at IGN/exp/experiment_classification.py:136.

    Parameters become views into `flat_param` (same trick as the gradient bucket), so the model, its state_dict and
    checkpoints are unaffected.  Needs the bucket (its flat gradient is the kernel's input).  A torch Optimizer
    subclass, so lr schedulers (CosineAnnealingWarmRestarts at :137) attach to it; the learning rate is read from
    ``param_groups[0]['lr']`` at every step.
    """

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        from . import _lib
        ps = bucket.params
        if not ps[0].is_cuda:
            raise _lib.IgnError("FlatAdam runs on the GPU only")
        super().__init__(ps, dict(lr=lr, betas=betas, eps=eps))
        self._lib = _lib
        self.bucket, self.lr, self.betas, self.eps = bucket, lr, betas, eps
        self.step_count = 0
        self.flat_param = torch.zeros_like(bucket.flat_grad)
        with torch.no_grad():
            for p, off in zip(ps, bucket.offsets):
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p)
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        # capturable: the step count lives on the device (ign_adam_step_dev), so the step can sit inside a hipGraph
        self.capturable = bool(capturable)
        self.step_dev = torch.zeros(1, device=self.flat_param.device, dtype=torch.int32) if capturable else None
        self.bc_dev = torch.zeros(2, device=self.flat_param.device, dtype=torch.float32) if capturable else None

    def make_capturable(self):
        """Move the step count to the device (ign_adam_step_dev) so that the optimizer step can be captured into a hipGraph;
        the trajectory is unchanged (same update rule, the count continues where the host count stands)."""
        if not self.capturable:
            dev = self.flat_param.device
            self.step_dev = torch.full((1,), self.step_count, device=dev, dtype=torch.int32)
            self.bc_dev = torch.zeros(2, device=dev, dtype=torch.float32)
            self.capturable = True

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        loss = closure() if closure is not None else None
        self.bucket.gather()
        self.step_count += 1
        g = self.param_groups[0]
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        self._lib.PARAM_GENERATION[0] += 1           # the kernel rewrites the parameters through raw pointers
        if self.capturable:
            self._lib.check(self._lib.lib().ign_adam_step_dev(
                ptr(self.flat_param), ptr(self.bucket.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                self.flat_param.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"], ptr(self.step_dev), ptr(self.bc_dev),
                self._lib.stream()), "ign_adam_step_dev")
            return loss
        self._lib.check(self._lib.lib().ign_adam_step(
            ptr(self.flat_param), ptr(self.bucket.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
            self.flat_param.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.step_count,
            self._lib.stream()), "ign_adam_step")
        return loss

    def zero_grad(self, set_to_none=False):
        self.bucket.zero_grad()
