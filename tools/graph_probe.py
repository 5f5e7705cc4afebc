"""GPU-box probe: capture the eval forward in a HIP graph (torch.cuda.CUDAGraph) and compare latency with eager."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import UNet_Nested  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNet_Nested(in_channels=1, n_classes=4, feature_scale=1).to(dev).eval()
for b, size in ((1, 256), (4, 256), (32, 256), (1, 64)):
    x = torch.randn(b, 1, size, size, device=dev)
    with torch.no_grad():
        for _ in range(3):
            ref = model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20
        static_x = x.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                model(static_x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = model(static_x)
        static_x.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        err = max((o - r).abs().max().item() for o, r in zip(outs, ref))
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 50
    print("B=%d %dx%d: eager %.3f ms, graph %.3f ms, max |diff| %.2e" % (b, size, size, eager * 1e3, graph * 1e3, err))
