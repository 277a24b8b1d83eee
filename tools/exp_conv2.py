import time, torch, torch.nn as nn
dev = "cuda:0"; N = 65536; c = 81
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
layers, cin = [], 2
for _ in range(4):
    layers += [nn.Conv2d(cin, 56, 3, padding=1), nn.BatchNorm2d(56), nn.ReLU()]; cin = 56
body = nn.Sequential(*layers).to(dev).eval().to(memory_format=torch.channels_last)
fused = nn.Sequential(*[torch.nn.utils.fusion.fuse_conv_bn_eval(body[i], body[i + 1]) if isinstance(body[i], nn.Conv2d) else body[i]
                        for i in range(len(body)) if not isinstance(body[i], nn.BatchNorm2d)]).to(memory_format=torch.channels_last)
def head(ch, out):
    return nn.Sequential(nn.Conv2d(56, ch, 1), nn.Flatten(), nn.LayerNorm(ch * c), nn.ReLU(), nn.Linear(ch * c, 128),
                         nn.LayerNorm(128), nn.ReLU(), nn.Linear(128, out)).to(dev).eval()
actor, critic = head(2, c), head(1, 1)
x = torch.randint(0, 2, (N, 2, 9, 9), device=dev).float().contiguous(memory_format=torch.channels_last)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    print("body with BN  :", timeit(lambda: body(x)), "ms")
    print("body BN folded:", timeit(lambda: fused(x)), "ms")
    f = fused(x)
    print("actor head    :", timeit(lambda: actor(f)), "ms")
    print("critic head   :", timeit(lambda: critic(f)), "ms")
    lg = actor(f).float()
    print("Categorical   :", timeit(lambda: torch.distributions.Categorical(logits=lg, validate_args=False).logits), "ms")
