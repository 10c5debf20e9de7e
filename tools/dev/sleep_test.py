import torch, time
torch.cuda.synchronize()
for c in (10**5, 10**6, 10**7):
    torch.cuda.synchronize(); t0=time.perf_counter(); torch.cuda._sleep(c); torch.cuda.synchronize(); print(c, (time.perf_counter()-t0)*1e3, "ms")
