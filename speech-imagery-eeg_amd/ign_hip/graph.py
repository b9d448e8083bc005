"""Whole-step hipGraph capture for launch-bound regimes (small batches: the UEA runs of run_uea.sh use B=32, where the
~130 kernel launches of an IGN step cost more host time than the kernels take on the GPU).

``GraphedTrainStep`` captures forward + loss + backward + gradient all-reduce (if any) + Adam of ONE step into a
``torch.cuda.CUDAGraph`` (a hipGraph on ROCm) over static input buffers and replays it per batch: one host call per step.
Everything in the library is capture-safe: kernels are enqueued on the current stream, the two expert streams fork and
join through events, no launch depends on host state that changes between steps (the optimizer must be
``FlatAdam(..., capturable=True)``: device-side step count).  Shapes must stay fixed; a ragged last batch is run eagerly.
"""
import torch


class GraphedTrainStep:
    def __init__(self, step_fn, example_inputs, warmup=3):
        """step_fn(*static_inputs) -> loss tensor; runs forward, backward, optimizer step, zero_grad."""
        self.step_fn = step_fn
        self.static = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up off the default stream, as torch.cuda.graphs asks
            for _ in range(warmup):
                self.step_fn(*self.static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self.step_fn(*self.static)

    def __call__(self, *inputs):
        if any(i.shape != s.shape for i, s in zip(inputs, self.static)):
            return self.step_fn(*inputs)                    # ragged batch: eager
        for s, i in zip(self.static, inputs):
            s.copy_(i, non_blocking=True)
        self.graph.replay()
        return self.loss
