import torch, time
n = 2048*2048*50
a = torch.zeros(n, dtype=torch.float64, device='cuda'); b = torch.ones(n, dtype=torch.float64, device='cuda'); c = torch.ones(n, dtype=torch.float64, device='cuda')
def t(f, nb, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{nb/dt/1e12:.2f} TB/s  ({dt*1e3:.3f} ms)")
print("copy (1R+1W)"); t(lambda: a.copy_(b), 2*n*8)
print("fill (1W)"); t(lambda: a.fill_(1.0), n*8)
print("add (2R+1W)"); t(lambda: torch.add(b, c, out=a), 3*n*8)
print("sum (1R)"); t(lambda: b.sum(), n*8)
