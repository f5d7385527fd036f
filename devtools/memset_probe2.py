"""Which zero-fills survive HIP-graph replays?  torch.zeros / zero_ / fill_ at several sizes, captured once, replayed 5 times on
buffers dirtied in between."""
import json, sys, torch
dev = torch.device("cuda:0")
out = {}
for n in (1, 3, 16, 1000, 4096, 1 << 20, 5_000_000):
    for kind in ("zero_", "fill_0", "zeros_like_then_copy", "mul0"):
        x = torch.ones(n, device=dev)
        y = torch.ones(n, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())

        def op():
            if kind == "zero_":
                x.zero_()
            elif kind == "fill_0":
                x.fill_(0.0)
            elif kind == "zeros_like_then_copy":
                x.copy_(torch.zeros_like(y))
            else:
                x.mul_(0.0)

        with torch.cuda.stream(side):
            op()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            op()
        bad = 0
        for it in range(5):
            x.fill_(1.0)
            g.replay()
            torch.cuda.synchronize()
            bad += int(float(x.abs().sum()) != 0.0)
        out[f"{kind}[{n}]"] = bad
print(json.dumps(out))
