import torch, time
n = 24*1024*1024
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
d = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(4)]
def run(k, reps=40):
    ss = [torch.cuda.Stream() for _ in range(k)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for r in range(reps):
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                d[i].copy_(h[i], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return reps * k * n / dt / 1e9
for k in (1, 2, 4):
    print(k, "streams:", round(run(k), 1), "GB/s H2D")
def run_d2h(k, reps=40):
    ss = [torch.cuda.Stream() for _ in range(k)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for r in range(reps):
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                h[i].copy_(d[i], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return reps * k * n / dt / 1e9
print("D2H 1 stream", round(run_d2h(1), 1))
