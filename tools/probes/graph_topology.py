"""Is the captured training step the plain chain graph.py assumes?

Captures GraphedTrainStep for the parametrisation that disagreed with the eager step in the driver's round-4 run (bf16
storage, bilinear up path, feature_scale 2, eager SGD; GPUTEST_r04.json) and for the captured-optimizer flavours, has
hipGraphDebugDotPrint write nodes and edges (CUDAGraph.debug_dump) and reports: node count by kind, edge count, nodes
with more than one successor (forks) / predecessor (joins), roots and leaves.  A chain has edges = nodes - 1, one root,
one leaf, no fork.  Run on the GPU box:  python tools/probes/graph_topology.py [out_dir]
"""
import collections
import os
import re
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, UNet_Nested  # noqa: E402


def analyse(path):
    txt = open(path).read()
    nodes, edges = {}, []
    for m in re.finditer(r'^\s*"?([\w.:-]+)"?\s*\[(.*?)\];?\s*$', txt, re.M | re.S):
        name, attrs = m.group(1), m.group(2)
        if name in ("graph", "node", "edge"):
            continue
        lab = re.search(r'label\s*=\s*"(.*?)"', attrs, re.S)
        nodes[name] = (lab.group(1) if lab else "")[:120].replace("\n", " ")
    for m in re.finditer(r'^\s*"?([\w.:-]+)"?\s*->\s*"?([\w.:-]+)"?', txt, re.M):
        edges.append((m.group(1), m.group(2)))
        nodes.setdefault(m.group(1), "")
        nodes.setdefault(m.group(2), "")
    succ, pred = collections.Counter(a for a, _ in edges), collections.Counter(b for _, b in edges)
    kinds = collections.Counter()
    for lab in nodes.values():
        k = "kernel"
        low = lab.lower()
        for key in ("memset", "memcpy", "empty", "event", "host"):
            if key in low:
                k = key
        kinds[k] += 1
    forks = [n for n in nodes if succ[n] > 1]
    joins = [n for n in nodes if pred[n] > 1]
    roots = [n for n in nodes if pred[n] == 0]
    leaves = [n for n in nodes if succ[n] == 0]
    print("  nodes %d  edges %d  kinds %s" % (len(nodes), len(edges), dict(kinds)))
    print("  roots %d  leaves %d  forks %d  joins %d  -> %s" % (
        len(roots), len(leaves), len(forks), len(joins),
        "CHAIN" if (len(edges) == len(nodes) - 1 and len(roots) == 1 and len(leaves) == 1 and not forks) else "NOT a chain"))
    for n in (forks + joins)[:12]:
        print("    fork/join node %s: %s (succ %d, pred %d)" % (n, nodes[n], succ[n], pred[n]))
    for n in roots[:6]:
        print("    root %s: %s" % (n, nodes[n]))
    for n in leaves[:6]:
        print("    leaf %s: %s" % (n, nodes[n]))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/graph_topology"
    os.makedirs(out, exist_ok=True)
    dev = torch.device("cuda:0")
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    cases = [
        ("bf16-bilinear-sgd-eager", dict(in_channels=1, n_classes=4, feature_scale=2, is_deconv=False), True, "sgd", False),
        ("f32-adam-captured", dict(in_channels=1, n_classes=4, feature_scale=4), False, "adam", True),
        ("bf16-d5-adam-captured", dict(in_channels=3, n_classes=5, feature_scale=4, depth=5), True, "adam", True),
    ]
    from unet_nested4tiny_objects_keypoints_amd._lib import debug_switch
    variants = [("", {}, None)]
    if os.environ.get("TOPO_VARIANTS", "1") == "1":
        # the two round-4 workarounds, switched OFF one at a time: what does the captured graph look like then?
        variants += [("+threaded-backward", {"_threaded_backward": True}, None), ("+memset-nodes", {}, ("MEMSET_NODES", 1))]
    for (name, ctor, bf16, kind, cap), (vname, kw, switch) in [(c, v) for c in cases for v in variants]:
        name = name + vname
        torch.manual_seed(81)
        m = UNet_Nested(**ctor).to(dev).train()
        if bf16:
            m.set_activation_dtype(torch.bfloat16)
        m.drop_out.p = 0.0
        opt = (torch.optim.SGD(m.parameters(), lr=2e-3, momentum=0.9) if kind == "sgd"
               else torch.optim.Adam(m.parameters(), lr=1e-3, fused=True, capturable=True))
        x = torch.randn(2, ctor["in_channels"], 64, 64, device=dev)
        t = torch.rand(2, ctor["n_classes"], 64, 64, device=dev)
        path = os.path.join(out, name + ".dot")
        import contextlib
        with (debug_switch(*switch) if switch else contextlib.nullcontext()):
            step = GraphedTrainStep(m, opt, crit, x, t, capture_optimizer=cap, debug_dot=path, **kw)
        print(name, "hipGraphGetNodes / hipGraphGetEdges:", step.topology)
        print(name, "->", path, "(%d bytes)" % (os.path.getsize(path) if os.path.exists(path) else -1))
        if os.path.exists(path):
            analyse(path)
        # a few replays, each from an idle device (where the round-4 memset symptom showed); lr is small, so a loss
        # that jumps or a non-finite gradient is visible at a glance
        losses = []
        for _ in range(4):
            torch.cuda.synchronize()
            step(x, t)
            torch.cuda.synchronize()
            losses.append(float(step._loss))
        worst = max(float(p.grad.abs().max()) for p in m.parameters())
        print("  replays: losses %s, largest |grad| %.3e" % (["%.4f" % v for v in losses], worst))


if __name__ == "__main__":
    main()
