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
