"""hipGraph capture of the inference forward.

The forward of one batch is ~60 short launches (12 rspmm + fix-ups, 12 fused updates, two batched
relation-projection GEMMs, index glue); on a static graph with a fixed batch shape the launch sequence never
changes, so it is captured once into a HIP graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture /
hipGraphLaunch) and replayed: one host call per forward instead of ~60.  Our kernels are enqueued on
torch's current stream, which is the capturing stream during capture; plans, scratch buffers and the
LDS opt-in are created by the eager warm-up runs, so nothing allocates inside the captured region.
"""
import torch


class GraphedForward(object):
    """score = GraphedForward(model, data, example_batch)(batch) for batches of example_batch's shape."""

    def __init__(self, model, data, example_batch, warmup=3):
        assert example_batch.is_cuda, "graph capture needs GPU tensors"
        self.model = model
        self.data = data
        self.static_batch = example_batch.clone()
        model.eval()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):
                model(data, self.static_batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(data, self.static_batch)
        self.valid = getattr(model.entity_model, "_pending_valid", None) if hasattr(model, "entity_model") else None

    def __call__(self, batch, check=False):
        if batch.shape != self.static_batch.shape:
            raise ValueError("GraphedForward was captured for batch shape %s, got %s"
                             % (tuple(self.static_batch.shape), tuple(batch.shape)))
        self.static_batch.copy_(batch, non_blocking=True)
        self.graph.replay()
        if check and self.valid is not None:
            assert bool(self.valid.all()), "every row of `batch` must share its head (or tail) and its relation (models.py:196-197)"
        return self.static_out
