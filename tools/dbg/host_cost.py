import time, torch
t0 = time.perf_counter(); p = [torch.randperm(2048) < 2048 for _ in range(1024)]; t1 = time.perf_counter()
print("1024 randperm(2048): %.1f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); n = [torch.randn((128, 20, 256)).transpose(1, 2) for _ in range(6 * 8)]; t1 = time.perf_counter()
print("48 x randn(128,20,256): %.1f ms" % ((t1 - t0) * 1e3))
x = n[0].contiguous()
torch.cuda.synchronize()
t0 = time.perf_counter(); y = [a.to("cuda", torch.float32).contiguous() for a in n]; torch.cuda.synchronize(); t1 = time.perf_counter()
print("48 H2D of 2.6 MB: %.1f ms" % ((t1 - t0) * 1e3))
