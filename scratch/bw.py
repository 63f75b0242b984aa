import torch, time
def bw(f, nbytes, n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return nbytes*n/(time.perf_counter()-t)/1e9
N=2**28  # 2 GiB of doubles
a=torch.randn(N, dtype=torch.float64, device='cuda'); b=torch.empty_like(a)
print('copy  GB/s (r+w)', bw(lambda: b.copy_(a), 16*N))
print('read  GB/s (sum)', bw(lambda: a.sum(), 8*N))
print('write GB/s (fill)', bw(lambda: b.fill_(1.0), 8*N))
print('rmw   GB/s (add_)', bw(lambda: a.add_(1.0), 16*N))
