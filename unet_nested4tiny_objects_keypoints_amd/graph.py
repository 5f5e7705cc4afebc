"""One whole training step replayed from a HIP graph.  EXPERIMENTAL: nothing on a default path uses it.

The reference's loop body (trainer/trainer.py:114-136: zero_grad -> forward -> criterion on every head -> mean ->
backward -> optimizer.step()) is ~200-280 kernel launches per step on this path.  Eager PyTorch needs 4.5-7 ms of host
time to enqueue them; the bf16 steps of BASELINE configs[3] / configs[4] take 7.2 / 10.9 ms on the device, so the host is
never far ahead.  ``GraphedTrainStep`` captures the step once for fixed tensor shapes (torch.cuda.CUDAGraph = hipGraph)
and replays it with one call: the launches, their order and their arguments are those of ``train_step`` -- the same
kernels do the same work -- only the enqueueing is gone (host 7 ms -> 0.15 ms per step; the DEVICE time of these steps
does not change, they are device-bound: DESIGN.md section 8).

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

Capture discipline (round 5, after the driver's run of round 4 saw a replay and the eager step disagree):
  * warm-up and capture run on ONE private stream (``torch.cuda.graph(g, stream=side)``): the gradient accumulators
    PyTorch creates for the parameters are bound to the stream they are first used on, and an accumulator bound to
    another stream than the producer of its gradient makes the autograd engine put event hops into the capture -- a
    fork / join DAG instead of a chain.  PyTorch's warning about that ("AccumulateGrad node's stream does not match")
    is an ERROR inside ``__init__``;
  * the capture starts with no autograd state from the warm-up (``p.grad = None``, no output or loss of a warm-up
    step alive) and the step hands out DETACHED static outputs and loss, so no autograd graph -- and no accumulator
    bound to the capture stream -- outlives the capture; eager passes of the same model afterwards create their own;
  * backward is captured with single-threaded autograd (every captured launch is enqueued by this thread);
  * ``check_topology=True`` keeps the hipGraph_t (``CUDAGraph(keep_graph=True)``) and reads its nodes and edges back
    through hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType (``self.topology``; ``debug_dot`` also has
    hipGraphDebugDotPrint write them out): tests/test_gpu_graph.py asserts that what was captured is the chain this
    file assumes -- one root, one leaf, no node with two successors or two predecessors.

The warm-up passes that size the allocator pools and record the weight-image jobs are real training steps on the
example batch; by default their effect is undone: parameters and buffers are restored in place, and the optimizer's
state is put back from a deep copy taken before the warm-up -- tensors in place, Python numbers by value (the
reference's tools/optimizers/adamw.py:62,76 and adabound.py:78,92 keep ``state['step']`` as an int), entries the
warm-up created are removed (``capture_optimizer=False``) or zeroed in place (captured optimizer: the graph holds their
addresses).

Not supported: a model with a data-parallel averager attached (the gradient all-reduce runs on a side stream with its
own events: use ``train_step`` there), CPU tensors (this path has no CPU fallback).
"""
from __future__ import annotations

import collections
import copy
import ctypes
import os
import warnings
from typing import Optional

import torch

from .step import loss_and_backward

_HIP_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
                   7: "event_record", 8: "ext_sem_signal", 9: "ext_sem_wait", 10: "mem_alloc", 11: "mem_free",
                   12: "memcpy_from_symbol", 13: "memcpy_to_symbol"}


def _loaded_hip_runtime():
    """The libamdhip64 image THIS process captured the graph with (the one torch loaded), opened by its exact path with
    RTLD_NOLOAD: the image ships two HIP runtimes (torch/lib and /opt/rocm/lib), and a hipGraph_t handed to the other
    one is a foreign pointer."""
    path = None
    with open("/proc/self/maps") as f:
        for ln in f:
            if "libamdhip64.so" in ln:
                path = ln.split()[-1]
                break
    if path is None:
        raise RuntimeError("no libamdhip64.so is loaded in this process (is torch's HIP runtime initialised?)")
    return ctypes.CDLL(path, mode=getattr(os, "RTLD_NOLOAD", 4) | getattr(os, "RTLD_NOW", 2))


def graph_topology(raw_graph: int, dot_path: Optional[str] = None) -> dict:
    """Nodes and edges of a hipGraph_t (the integer torch.cuda.CUDAGraph.raw_cuda_graph() returns), read back through the
    HIP runtime this process already has loaded.  ``chain`` is True when the graph is one path: a single root, a single
    leaf, edges = nodes - 1, no fork, no join."""
    hip = _loaded_hip_runtime()
    g = ctypes.c_void_p(raw_graph)

    def ok(status, what):
        if status != 0:
            raise RuntimeError("%s failed with hipError %d" % (what, status))
    n = ctypes.c_size_t(0)
    ok(hip.hipGraphGetNodes(g, None, ctypes.byref(n)), "hipGraphGetNodes (count)")
    nodes = (ctypes.c_void_p * max(1, n.value))()
    ok(hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)), "hipGraphGetNodes")
    e = ctypes.c_size_t(0)
    ok(hip.hipGraphGetEdges(g, None, None, ctypes.byref(e)), "hipGraphGetEdges (count)")
    src, dst = (ctypes.c_void_p * max(1, e.value))(), (ctypes.c_void_p * max(1, e.value))()
    ok(hip.hipGraphGetEdges(g, src, dst, ctypes.byref(e)), "hipGraphGetEdges")
    kinds = collections.Counter()
    kind_of = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        ok(hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)), "hipGraphNodeGetType")
        kind_of[nodes[i]] = _HIP_NODE_TYPES.get(t.value, "type%d" % t.value)
        kinds[kind_of[nodes[i]]] += 1
    succ, pred = collections.Counter(src[i] for i in range(e.value)), collections.Counter(dst[i] for i in range(e.value))
    ids = [nodes[i] for i in range(n.value)]
    forks = [kind_of[v] for v in ids if succ[v] > 1]
    joins = [kind_of[v] for v in ids if pred[v] > 1]
    roots = [kind_of[v] for v in ids if pred[v] == 0]
    leaves = [kind_of[v] for v in ids if succ[v] == 0]
    if dot_path is not None:
        hip.hipGraphDebugDotPrint(g, dot_path.encode(), ctypes.c_uint(0))
    return {"nodes": n.value, "edges": e.value, "kinds": dict(kinds), "roots": roots, "leaves": leaves, "forks": forks,
            "joins": joins,
            "chain": bool(n.value >= 1 and e.value == n.value - 1 and len(roots) == 1 and len(leaves) == 1 and not forks and not joins)}


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer, criterion, x, target); outputs, loss = step(x, target)

    ``outputs`` / ``loss`` are the graph's static tensors (detached): the next call overwrites them (clone what must
    live longer).  capture_optimizer=True puts ``optimizer.step()`` into the graph; it needs an optimizer whose step
    makes no host decision that depends on device data (torch.optim.Adam / AdamW with ``capturable=True``, SGD)."""

    def __init__(self, model, optimizer, criterion, inputs: torch.Tensor, target: torch.Tensor, warmup: int = 3,
                 capture_optimizer: bool = False, restore_state: bool = True, check_topology: bool = False,
                 debug_dot: Optional[str] = None, _threaded_backward: bool = False):
        if not inputs.is_cuda or not target.is_cuda:
            raise RuntimeError("GraphedTrainStep needs GPU tensors: this path has no CPU fallback")
        if capture_optimizer:
            # The state tensors a captured optimizer created in the warm-up are graph addresses; they are put back to ZERO
            # before the first replay (_restore).  That is the optimizer's fresh state only when "zeros, then one update"
            # equals its first step: Adam / AdamW, SGD with dampening 0 (momentum_buffer = grad on the first step).
            for grp in optimizer.param_groups:
                if grp.get("dampening", 0) != 0:
                    raise ValueError("capture_optimizer=True needs an optimizer whose zeroed state is its fresh state: "
                                     "SGD with dampening != 0 initialises momentum_buffer = grad, not (1 - dampening) * grad")
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
                     copy.deepcopy(optimizer.state_dict()), {id(p) for p in optimizer.state})
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        check_topology = check_topology or debug_dot is not None
        self._graph = torch.cuda.CUDAGraph(keep_graph=True) if check_topology else torch.cuda.CUDAGraph()
        self.topology = None
        with warnings.catch_warnings():
            warnings.filterwarnings("error", message=".*AccumulateGrad node's stream does not match.*")
            # Backward is captured from THIS thread (PyTorch would run it on a per-device worker thread).
            # (_threaded_backward: tools/probes/graph_topology.py only -- what PyTorch's worker thread does to the capture)
            with torch.autograd.set_multithreading_enabled(bool(_threaded_backward)):
                with torch.cuda.stream(side):
                    for _ in range(max(1, warmup)):   # records the weight-image jobs, sizes the pools, creates optimizer state
                        self._new_seed()
                        self._body(True)
                    optimizer.zero_grad(set_to_none=True)    # the capture starts without autograd state of the warm-up
                side.synchronize()
                # the captured launches carry raw pointers into the model's weight-image plan: keep it alive and un-evicted
                self._plan = model.__dict__.get("_pack_plan")
                if self._plan is not None:
                    self._plan.pin()
                with torch.cuda.graph(self._graph, stream=side):
                    outs, loss = self._body(self.capture_optimizer)
                    outs = tuple(o.detach() for o in outs) if isinstance(outs, tuple) else outs.detach()
                    loss = loss.detach()
        torch.cuda.current_stream(dev).wait_stream(side)
        if check_topology:
            self.topology = graph_topology(self._graph.raw_cuda_graph(), debug_dot)
            self._graph.instantiate()
        # eager passes of the model (evaluation, instrumented steps) must draw their own seeds again: the captured head
        # launches hold the address of the device word themselves
        model._dropout_seed_dev = None
        self._held = None if self._plan is None else (dict(self._plan._tables), [e.image for e in self._plan.entries.values()])
        self._outs, self._loss = outs, loss
        self._grads = [(p, p.grad) for p in params if p.grad is not None]
        self._param_ptrs = [p.data_ptr() for p in params]
        if saved is not None:
            self._restore(params, saved)
        torch.cuda.synchronize(dev)

    def _restore(self, params, saved):
        """Undo the warm-up: parameters, buffers, optimizer state as they were before ``__init__``."""
        with torch.no_grad():
            for p, v in zip(params, saved[0]):
                p.copy_(v)
            for b, v in zip(self.model.buffers(), saved[1]):
                b.copy_(v)
            before, had = saved[2], saved[3]
            ids = [p for grp in self.optimizer.param_groups for p in grp["params"]]
            old_state = before["state"]                       # keyed by the parameter's index in param_groups order
            for idx, p in enumerate(ids):
                st = self.optimizer.state.get(p)
                if st is None:
                    continue
                if id(p) not in had and not self.capture_optimizer:
                    del self.optimizer.state[p]               # created by the warm-up: the first real step creates it again
                    continue
                old = old_state.get(idx, {})
                for k in list(st.keys()):
                    v = st[k]
                    if torch.is_tensor(v):
                        if k in old and torch.is_tensor(old[k]):
                            v.copy_(old[k])
                        else:
                            v.zero_()                         # captured optimizer: its state tensors are graph addresses
                    elif k in old:
                        st[k] = copy.deepcopy(old[k])
                    elif isinstance(v, (int, float)):
                        st[k] = type(v)(0)
        if self._plan is not None:
            self._plan.invalidate()

    def _new_seed(self):
        # CPU generator, as the eager path draws its seed (engine._dropout_config); fill_ takes it as a launch argument
        self._seed.fill_(int(torch.randint(0, 2 ** 62, (1,)).item()))

    def _body(self, with_optimizer: bool):
        """train_step's sequence (step.py) on the static tensors"""
        self.optimizer.zero_grad(set_to_none=True)
        outputs = self.model(self._x)
        avgloss = loss_and_backward(self.criterion, outputs, self._t)
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
