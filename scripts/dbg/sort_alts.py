import torch, time
dev = torch.device("cuda", 0)
x = torch.randn(100_000, device=dev)
def timed(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("sort desc stable        ", timed(lambda: torch.sort(x, descending=True, stable=True)))
print("sort desc               ", timed(lambda: torch.sort(x, descending=True)))
print("argsort desc            ", timed(lambda: torch.argsort(x, descending=True)))
print("topk 30000 sorted       ", timed(lambda: torch.topk(x, 30000, sorted=True)))
print("topk 30000 unsorted     ", timed(lambda: torch.topk(x, 30000, sorted=False)))
print("topk 1000 sorted        ", timed(lambda: torch.topk(x, 1000, sorted=True)))
print("kthvalue                ", timed(lambda: torch.kthvalue(x, 70000)))
xi = x.view(torch.int32)
print("sort int32              ", timed(lambda: torch.sort(xi, descending=True)))
xh = x.to(torch.bfloat16)
print("sort bf16               ", timed(lambda: torch.sort(xh, descending=True)))
x64 = (x.view(torch.int32).to(torch.int64) << 20) + torch.arange(100_000, device=dev)
print("sort int64 keys (no idx)", timed(lambda: torch.sort(x64)[0]))
for n in (1_000_000,):
    x = torch.randn(n, device=dev)
    print(n, "sort desc stable", timed(lambda: torch.sort(x, descending=True, stable=True), 50))
    print(n, "kthvalue        ", timed(lambda: torch.kthvalue(x, int(n * 0.7)), 50))
    print(n, "topk 30% sorted  ", timed(lambda: torch.topk(x, int(n * 0.3), sorted=True), 50))
    print(n, "topk 30% unsorted", timed(lambda: torch.topk(x, int(n * 0.3), sorted=False), 50))
    xs = torch.sort(x, descending=True).values
    print(n, "sort + index (threshold via sort)", timed(lambda: torch.sort(x, descending=True).values[int(n * 0.3) - 1], 50))
