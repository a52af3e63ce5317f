"""what a large device allocation costs by size (hipMalloc through torch's allocator, cache emptied between): tools/alloc_probe.py"""
import time, torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for gb in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64):
    t0 = time.perf_counter(); x = torch.empty(int(gb*1e9), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    x[::4096].fill_(1); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%3d GB: allocate %.3f s, touch every page %.3f s" % (gb, t1-t0, t2-t1), flush=True); del x; torch.cuda.empty_cache()
t0 = time.perf_counter(); xs = [torch.empty(int(12e9), dtype=torch.uint8, device="cuda") for _ in range(4)]; torch.cuda.synchronize(); t1 = time.perf_counter()
print("4 x 12 GB: allocate %.3f s" % (t1-t0))
