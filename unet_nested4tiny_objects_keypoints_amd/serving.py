"""Eval forward replayed from a HIP graph (small-batch serving).

At batch 1 the eval forward of UNet_Nested is about sixty launches of a few microseconds each: eager PyTorch spends
more time enqueueing them than the device spends running them (1.14 ms vs 0.72 ms per 256x256 image on MI355X).
``GraphedForward`` captures one eval forward for a fixed input shape (torch.cuda.CUDAGraph = hipGraph) and replays it;
from batch 4 on the device is the bound and the graph changes nothing (tools/graph_probe.py), so this is a latency
tool, not a throughput one.  The reference serves with eager PyTorch (trainer/trainer.py:141-180 validation loop); this
is the MI355X-side addition for that loop's forward.

The weight images are rebuilt by a launch INSIDE the graph (ops.PackPlan), so parameter updates between replays are
seen; call ``model.freeze_weight_images()`` before capturing to drop that launch for frozen weights.
"""
from typing import Tuple

import torch


class GraphedForward:
    """outs = GraphedForward(model, example)(x) -- eval forward of ``model`` for inputs shaped like ``example``.

    The returned tensors are the graph's static outputs: they are overwritten by the next call (clone what must live
    longer)."""

    def __init__(self, model: torch.nn.Module, example: torch.Tensor, warmup: int = 3):
        if not example.is_cuda:
            raise RuntimeError("GraphedForward needs a GPU input: this path has no CPU fallback")
        if model.training:
            raise RuntimeError("GraphedForward captures the eval forward: call model.eval() first")
        self.model = model
        self._x = example.clone()
        side = torch.cuda.Stream(device=example.device)   # warm-up AND capture run on this one stream (as graph.py)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(max(1, warmup)):  # first passes record the weight-image jobs and size the allocator pools
                model(self._x)
        side.synchronize()
        # The captured launches carry raw pointers into the model's weight-image plan (the images and the device job
        # table of the in-graph pack launch): pin the plan so that neither is evicted / freed while this graph lives,
        # and remember what was captured -- a replay after the plan changed shape (set_activation_dtype, a differently
        # structured eager call) still finds its own table and images alive.
        self._plan = model.__dict__.get("_pack_plan")
        if self._plan is not None:
            self._plan.pin()
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph, stream=side):
            outs = model(self._x)
        torch.cuda.current_stream(example.device).wait_stream(side)
        self._held = None
        if self._plan is not None:
            self._held = (self._plan._tables.get("fwd"), [e.image for e in self._plan.entries.values()])
        self._param_ptrs = [p.data_ptr() for p in model.parameters()]
        self._outs: Tuple[torch.Tensor, ...] = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
        self._single = not isinstance(outs, (tuple, list))

    def __call__(self, x: torch.Tensor):
        if x.shape != self._x.shape or x.dtype != self._x.dtype or x.device != self._x.device:
            raise ValueError("GraphedForward was captured for %s %s on %s, got %s %s on %s" % (
                tuple(self._x.shape), self._x.dtype, self._x.device, tuple(x.shape), x.dtype, x.device))
        if self.model.training:
            raise RuntimeError("the captured graph is the eval forward; the model is in training mode")
        if [p.data_ptr() for p in self.model.parameters()] != self._param_ptrs:
            raise RuntimeError("the model's parameters moved (.to() / load with assign) since the graph was captured: "
                               "build a new GraphedForward")
        self._x.copy_(x)
        self._graph.replay()
        return self._outs[0] if self._single else self._outs

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan is not None:
            plan.unpin()
            self._plan = None
