"""Host -> device double buffering for the training loop.

Once the step runs in ~14 ms the synchronous ``.to(device)`` of IGN/exp/experiment_classification.py:315-317 and the
per-item CPU normalisation in the DataLoader workers become the bottleneck (a 256-sample CHISCO batch is 125 MB: ~2 ms
over PCIe Gen5, and ~0.5 ms of CPU z-scoring per sample).  ``DevicePrefetcher`` wraps any DataLoader of the batch contract
``(X, y, mask)``: batch i+1 is pinned and copied on a side HIP stream while batch i trains; an optional ``transform`` runs
on that stream too -- for the CHISCO shards the loader ships RAW (B,C,T) microvolt tensors and the per-sample
standardisation + transpose happen on the GPU (``ops.standardise_nct_to_btc``).  On a CPU device it is a pass-through.
"""
import torch


def standardise_raw_batch(batch):
    """transform for loaders that yield raw (X[B,C,T], y, None): -> (X[B,T,C] standardised, y, mask ones[B,T])."""
    from ign_hip import ops
    x, y, _ = batch
    xs = ops.standardise_nct_to_btc(x.float())
    return xs, y, torch.ones(xs.shape[0], xs.shape[1], device=xs.device, dtype=torch.bool)


class DevicePrefetcher:
    def __init__(self, loader, device, transform=None, depth=2):
        self.loader, self.device, self.transform, self.depth = loader, torch.device(device), transform, max(1, depth)
        self.dataset = getattr(loader, "dataset", None)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch, stream):
        def put(t):
            if not torch.is_tensor(t):
                return t
            if not t.is_pinned():
                t = t.pin_memory()
            return t.to(self.device, non_blocking=True)
        with torch.cuda.stream(stream):
            out = tuple(put(t) for t in batch)
            if self.transform is not None:
                out = self.transform(out)
            ev = torch.cuda.Event()
            ev.record(stream)
        return out, ev

    def __iter__(self):
        if self.device.type != "cuda":
            for batch in self.loader:
                yield self.transform(batch) if self.transform is not None else batch
            return
        stream = torch.cuda.Stream(self.device)
        queue = []
        it = iter(self.loader)
        done = False
        while True:
            while not done and len(queue) < self.depth:
                try:
                    queue.append(self._stage(next(it), stream))
                except StopIteration:
                    done = True
            if not queue:
                return
            out, ev = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in out:
                if torch.is_tensor(t):
                    t.record_stream(cur)        # allocated on the copy stream, consumed on the compute stream
            yield out


class TensorBatchLoader:
    """DataLoader stand-in for a data set whose samples are the rows of ONE in-memory tensor (``dataset.x`` (n, T, C) float32,
    ``dataset.y`` (n,)): the synthetic benchmark provider, and any fixed-length set after it has been read.

    A ``DataLoader`` fetches 256 items, stacks them (one 125 MB copy on one thread) and, with ``pin_memory``, copies the batch a
    second time -- at the CHISCO shape that is more host time than the 12 ms the step takes on the GPU.  Here a batch is ONE
    multi-threaded ``index_select`` straight into a pinned buffer (the caching host allocator hands the block back once the
    asynchronous copy that read it has completed), so the DevicePrefetcher's ``.to(device, non_blocking=True)`` is the only other
    copy.  Same batch contract as ``collate_fn`` (IGN/data_factory/uea.py:7-42): ``(X[B,T,C] float32, y[B,1], mask[B,T] bool)``;
    same iteration semantics as the DataLoader it replaces (fresh permutation per epoch from torch's global generator, a
    ``sampler`` yielding indices is honoured, ``drop_last=False``)."""

    def __init__(self, dataset, batch_size, shuffle=False, sampler=None, pin_memory=True):
        self.dataset, self.batch_size, self.shuffle, self.sampler = dataset, int(batch_size), bool(shuffle), sampler
        self.pin = bool(pin_memory) and torch.cuda.is_available()
        self.device_transform = None

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ds = self.dataset
        if self.sampler is not None:
            order = torch.as_tensor(list(iter(self.sampler)), dtype=torch.int64)
        elif self.shuffle:
            order = torch.randperm(len(ds))
        else:
            order = torch.arange(len(ds))
        T, C = ds.x.shape[1], ds.x.shape[2]
        for i in range(0, order.numel(), self.batch_size):
            idx = order[i:i + self.batch_size]
            k = idx.numel()
            X = torch.empty(k, T, C, dtype=torch.float32, pin_memory=self.pin)
            torch.index_select(ds.x, 0, idx, out=X)
            y = ds.y.index_select(0, idx).unsqueeze(1).to(torch.int8)
            yield X, y, torch.ones(k, T, dtype=torch.bool)
