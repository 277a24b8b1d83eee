"""How fast is the caller-side policy forward (MIOpen) at the config-3 shape?  (developer tool)"""
import time, torch, torch.nn as nn
dev = "cuda:0"
N = 65536
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
body = nn.Sequential(nn.Conv2d(2, 56, 3, padding=1), nn.ReLU(), nn.Conv2d(56, 56, 3, padding=1), nn.ReLU(),
                     nn.Conv2d(56, 56, 3, padding=1), nn.ReLU(), nn.Conv2d(56, 56, 3, padding=1), nn.ReLU()).to(dev).eval()
x = torch.randint(0, 2, (N, 2, 9, 9), device=dev).float()
with torch.no_grad():
    for name, dt, cl in (("fp32 nchw", None, False), ("bf16 nchw", torch.bfloat16, False), ("bf16 nhwc", torch.bfloat16, True),
                         ("fp16 nchw", torch.float16, False), ("fp16 nhwc", torch.float16, True)):
        b = body.to(memory_format=torch.channels_last) if cl else body.to(memory_format=torch.contiguous_format)
        xx = x.contiguous(memory_format=torch.channels_last) if cl else x
        def f():
            if dt is None:
                return b(xx)
            with torch.autocast("cuda", dtype=dt):
                return b(xx)
        print(f"{name}: {timeit(f):8.2f} ms / forward of 4 convs", flush=True)
    # conv as GEMM on an explicitly padded, unfolded input (bf16), in chunks
    w = [m.weight.reshape(56, -1).t().contiguous().to(torch.bfloat16) for m in body if isinstance(m, nn.Conv2d)]
    def gemm_conv():
        h = x.to(torch.bfloat16)
        for wi in w:
            cols = torch.nn.functional.unfold(h, 3, padding=1)           # (N, cin*9, 81)
            h = torch.relu(cols.transpose(1, 2).reshape(-1, cols.shape[1]) @ wi)   # (N*81, 56)
            h = h.reshape(N, 81, 56).transpose(1, 2).reshape(N, 56, 9, 9)
        return h
    print(f"unfold+GEMM bf16: {timeit(gemm_conv):8.2f} ms", flush=True)
