"""how the raw tallies of a batch of flux jobs reach the host: tools/d2h_probe.py"""
import time, torch
dev = torch.device('cuda', 0)
n = 48*3*70*128*128
buf = torch.rand(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for name, fn in (('float64 .cpu() (pageable)', lambda: buf.cpu()),
                 ('float32 on the device, .cpu() (pageable)', lambda: buf.to(torch.float32).cpu())):
    for r in range(2):
        t0 = time.perf_counter(); x = fn(); torch.cuda.synchronize(); dt = time.perf_counter()-t0
    print('%-46s %.3f s  (%.1f GB/s)' % (name, dt, x.numel()*x.element_size()/dt/1e9))
t0 = time.perf_counter(); pin = torch.empty(n, dtype=torch.float32, pin_memory=True); t1 = time.perf_counter()
print('pinning %.0f MB: %.3f s' % (pin.numel()*4/1e6, t1-t0))
for r in range(2):
    t0 = time.perf_counter(); pin.copy_(buf.to(torch.float32), non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter()-t0
print('float32 into pinned memory                     %.3f s  (%.1f GB/s)' % (dt, pin.numel()*4/dt/1e9))
