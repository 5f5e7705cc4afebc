"""One whole training step replayed from a HIP graph.

The reference's loop body (trainer/trainer.py:114-136: zero_grad -> forward -> criterion on every head -> mean ->
backward -> optimizer.step()) is ~200-280 kernel launches per step on this path.  Eager PyTorch needs 4.5-7 ms of host
time to enqueue them; the bf16 steps of BASELINE configs[3] / configs[4] take 7.2 / 10.9 ms on the device, so the host is
never far ahead, and wherever it falls behind -- the hand-over from the autograd thread to ``optimizer.step()`` is the
largest spot -- the device idles (0.5-0.7 ms per step in a rocprofv3 trace, profiles/r4/step_gaps_rocprofv3.txt).
``GraphedTrainStep`` captures the step once for fixed tensor shapes (torch.cuda.CUDAGraph = hipGraph) and replays it
with one call: the launches, their order and their arguments are those of ``train_step`` -- the same kernels do the
same work -- only the enqueueing is gone.

What a captured launch cannot take by value any more is handled explicitly:
  * inputs / targets are copied into the graph's static tensors before each replay;
  * dropout: the by-value seed of the head kernels is baked into the graph, so the varying part of the seed lives in a
    device word (``model._dropout_seed_dev``, ``seed_dev`` of unetpp_head_fwd / unetpp_head_bwd) that is rewritten from
    the CPU generator before every replay -- forward and backward of one step see the same value, every step a new one;
  * weight images are rebuilt by the pack launch INSIDE the graph (ops.PackPlan), so whatever updates the parameters
    (the captured optimizer step, or an eager optimizer working through ``p.data``) is seen by the next replay;
  * parameter gradients are the graph's static tensors; they are re-attached to ``p.grad`` after every replay, so an
    eager ``optimizer.step()`` (``capture_optimizer=False``: any optimizer, including the reference's own
    tools/optimizers/*) finds them even if something set ``p.grad`` to None in between.

The warm-up passes that size the allocator pools and record the weight-image jobs are real training steps on the
example batch; by default their effect is undone (parameters, buffers and optimizer state are restored in place,
optimizer state created during warm-up is zeroed) so that the first replay is the first step of the run.

Two properties of hipGraph on ROCm 7.2 shaped this file and csrc/ (both found with tools/probes/dbg_graph*.py, both silent):
launches captured from a second thread (PyTorch's autograd worker) leave the graph with a tail the launch stream does
not wait for -- backward is therefore captured with single-threaded autograd; and memset NODES are not reliably ordered
against neighbouring kernel nodes -- the library zeroes workspace rows with a kernel (pointwise.hip, zero_rows).

Not supported: a model with a data-parallel averager attached (the gradient all-reduce runs on a side stream with its
own events: use ``train_step`` there), CPU tensors (this path has no CPU fallback).
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer, criterion, x, target); outputs, loss = step(x, target)

    ``outputs`` / ``loss`` are the graph's static tensors: the next call overwrites them (clone what must live longer).
    capture_optimizer=True puts ``optimizer.step()`` into the graph; it needs an optimizer whose step makes no host
    decision that depends on device data (torch.optim.Adam / AdamW with ``capturable=True``, SGD)."""

    def __init__(self, model, optimizer, criterion, inputs: torch.Tensor, target: torch.Tensor, warmup: int = 3,
                 capture_optimizer: bool = False, restore_state: bool = True):
        if not inputs.is_cuda or not target.is_cuda:
            raise RuntimeError("GraphedTrainStep needs GPU tensors: this path has no CPU fallback")
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures a TRAINING step: call model.train() first")
        if getattr(model, "_grad_sink", None) is not None:
            raise RuntimeError("a data-parallel averager is attached to the model: its all-reduces are not captured; "
                               "use train_step")
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.capture_optimizer = bool(capture_optimizer)
        dev = inputs.device
        self._x, self._t = inputs.clone(), target.clone()
        self._seed = torch.zeros(1, dtype=torch.int64, device=dev)
        model._dropout_seed_dev = self._seed
        params = [p for p in model.parameters()]
        saved = None
        if restore_state:
            saved = ([p.detach().clone() for p in params], [b.detach().clone() for b in model.buffers()],
                     {id(t): t.detach().clone() for t in self._optimizer_tensors()})
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # Backward must be captured from THIS thread.  PyTorch runs the backward of device tensors on a per-device worker
        # thread; launches that another thread puts into a capturing stream do end up in the graph, but on ROCm 7.2 a
        # graph whose last nodes were captured from the second thread completes -- as far as the launch stream can tell
        # -- before those nodes have run: `stream.synchronize()` returned while the last weight-gradient kernels were
        # still writing (19 of 20 replays, tools/probes/dbg_graph3.py), and whatever the stream did next raced with them.
        # Single-threaded autograd keeps every captured launch on one thread and the graph a plain chain.
        with torch.cuda.stream(side), torch.autograd.set_multithreading_enabled(False):
            for _ in range(max(1, warmup)):   # records the weight-image jobs, sizes the pools, creates optimizer state
                self._new_seed()
                self._body(True)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # the captured launches carry raw pointers into the model's weight-image plan: keep it alive and un-evicted
        self._plan = model.__dict__.get("_pack_plan")
        if self._plan is not None:
            self._plan.pin()
        self._graph = torch.cuda.CUDAGraph()
        with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(self._graph):
            outs, loss = self._body(self.capture_optimizer)
        # eager passes of the model (evaluation, instrumented steps) must draw their own seeds again: the captured head
        # launches hold the address of the device word themselves
        model._dropout_seed_dev = None
        self._held = None if self._plan is None else (dict(self._plan._tables), [e.image for e in self._plan.entries.values()])
        self._outs, self._loss = outs, loss
        self._grads = [(p, p.grad) for p in params if p.grad is not None]
        self._param_ptrs = [p.data_ptr() for p in params]
        if saved is not None:
            with torch.no_grad():
                for p, v in zip(params, saved[0]):
                    p.copy_(v)
                for b, v in zip(model.buffers(), saved[1]):
                    b.copy_(v)
                for t in self._optimizer_tensors():
                    if id(t) in saved[2]:
                        t.copy_(saved[2][id(t)])
                    else:
                        t.zero_()   # state the warm-up created (moments, step counters): as before the first step
            if self._plan is not None:
                self._plan.invalidate()

    def _optimizer_tensors(self):
        out = []
        for st in self.optimizer.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    out.append(v)
        return out

    def _new_seed(self):
        # CPU generator, as the eager path draws its seed (engine._dropout_config); fill_ takes it as a launch argument
        self._seed.fill_(int(torch.randint(0, 2 ** 62, (1,)).item()))

    def _body(self, with_optimizer: bool):
        """train_step's sequence (step.py) on the static tensors"""
        self.optimizer.zero_grad(set_to_none=True)
        outputs = self.model(self._x)
        if isinstance(outputs, tuple):
            avgloss = 0
            for output in outputs:
                avgloss = avgloss + self.criterion(output, self._t)
            avgloss = 1.0 * avgloss / len(outputs)
        else:
            avgloss = self.criterion(outputs, self._t)
        avgloss.backward()
        if with_optimizer:
            self.optimizer.step()
        return outputs, avgloss

    def __call__(self, inputs: torch.Tensor, target: torch.Tensor):
        if inputs.shape != self._x.shape or inputs.dtype != self._x.dtype or inputs.device != self._x.device:
            raise ValueError("GraphedTrainStep was captured for inputs %s %s on %s, got %s %s on %s" % (
                tuple(self._x.shape), self._x.dtype, self._x.device, tuple(inputs.shape), inputs.dtype, inputs.device))
        if target.shape != self._t.shape or target.dtype != self._t.dtype or target.device != self._t.device:
            raise ValueError("GraphedTrainStep was captured for targets %s %s, got %s %s" % (
                tuple(self._t.shape), self._t.dtype, tuple(target.shape), target.dtype))
        if not self.model.training:
            raise RuntimeError("the captured graph is a training step; the model is in eval mode")
        if [p.data_ptr() for p in self.model.parameters()] != self._param_ptrs:
            raise RuntimeError("the model's parameters moved (.to() / load with assign) since the graph was captured: "
                               "build a new GraphedTrainStep")
        if inputs.data_ptr() != self._x.data_ptr():
            self._x.copy_(inputs)
        if target.data_ptr() != self._t.data_ptr():
            self._t.copy_(target)
        self._new_seed()
        self._graph.replay()
        for p, g in self._grads:
            p.grad = g
        if not self.capture_optimizer:
            self.optimizer.step()
        return self._outs, self._loss

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan is not None:
            plan.unpin()
            self._plan = None
