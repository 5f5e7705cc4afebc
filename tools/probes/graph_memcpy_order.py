"""Is a hipGraph launch ordered against the stream operations around it?  (torch only, no library code.)

GraphedTrainStep.__call__ does, on ONE stream: copy the new batch into the graph's static input tensors (Tensor.copy_ of
a contiguous same-dtype tensor = hipMemcpyAsync device-to-device), launch the graph, then run eager work that reads the
graph's results (the eager optimizer, .clone() of outputs).  The driver's round-4 failure was at the SECOND iteration of
the test loop -- the first one in which the copied batch differs from what the static tensors already hold -- with
parameters, gradients and buffers equal after the first: consistent with the head of the graph reading the static
tensors before the copy had landed.  This probe checks exactly that, without the library:

  head:  x.copy_(src_i)  [memcpy]  ->  replay (chain of K elementwise kernels, first one reads x)  ->  out must be f(i)
  tail:  replay  ->  z = out.clone() [memcpy]  /  z2 = out + 0 [kernel]  ->  both must be f(i)
in two regimes: from an idle device (synchronize before every iteration) and back to back; copies by memcpy and by kernel.
Prints the number of iterations with a stale result per (test, regime, copy kind).
"""
import sys

import torch


def build(n, k, dev):
    x = torch.zeros(n, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())

    def body():
        y = x * 1.0
        for _ in range(k):
            y = y + 1.0
        return y

    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return x, out, g


def main():
    dev = torch.device("cuda:0")
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    for n, k in ((8192, 4), (1 << 15, 40), (1 << 20, 40), (1 << 24, 8)):
        x, out, g = build(n, k, dev)
        srcs = [torch.full((n,), float(i % 97), device=dev) for i in range(8)]
        torch.cuda.synchronize()
        for regime in ("idle", "back-to-back"):
            for kind in ("memcpy", "kernel"):
                stale_head = stale_tail_memcpy = stale_tail_kernel = 0
                keep = []
                for i in range(iters):
                    src = srcs[i % 8]
                    want = float(i % 8 % 97) + k
                    if regime == "idle":
                        torch.cuda.synchronize()
                    if kind == "memcpy":
                        x.copy_(src)                       # contiguous, same dtype: hipMemcpyAsync D2D
                    else:
                        torch.maximum(src, src, out=x)     # the same bytes written by a kernel
                    g.replay()
                    z1 = out.clone()                       # memcpy behind the graph
                    z2 = out + 0.0                         # kernel behind the graph
                    keep.append((want, z1, z2))
                    if regime == "idle" or len(keep) >= 64:
                        torch.cuda.synchronize()
                        for w, a, b in keep:
                            # the head read stale x <=> both tails agree on a wrong value; a tail raced <=> they differ
                            va, vb = float(a[0]), float(b[0])
                            ok_a, ok_b = bool((a == w).all()), bool((b == w).all())
                            if not ok_a and not ok_b and va == vb:
                                stale_head += 1
                            else:
                                stale_tail_memcpy += int(not ok_a)
                                stale_tail_kernel += int(not ok_b)
                        keep = []
                print("n=%-9d k=%-3d %-13s input by %-6s: stale head %d, stale tail (memcpy reader) %d, (kernel reader) %d  of %d"
                      % (n, k, regime, kind, stale_head, stale_tail_memcpy, stale_tail_kernel, iters), flush=True)


if __name__ == "__main__":
    main()
