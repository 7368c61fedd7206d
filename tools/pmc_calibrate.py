"""Known-byte-count workloads for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (the guide
calibrates FETCH_SIZE only for wide coalesced streams).  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o c -- python tools/pmc_calibrate.py
and again with --pmc WRITE_SIZE; profiles/summarize.py reduces the counters per kernel."""
import torch
dev = torch.device('cuda', 0)
N = 1 << 30                                   # 1 Gi int32 = 4 GiB, beyond L2 + Infinity Cache
a = torch.ones(N, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
s = a.sum()                                   # streaming read of 4 GiB            -> reduce_kernel
torch.cuda.synchronize()
b = torch.empty(N // 2, dtype=torch.int64, device=dev)   # 4 GiB of 8-byte elements
b.fill_(3)                                    # streaming write of 4 GiB           -> fill kernel
torch.cuda.synchronize()
M = 1 << 26                                   # 64 Mi gathers of 8 bytes, each from its own random 128-B line
idx = (torch.randint(0, N // 2 // 16, (M,), device=dev, dtype=torch.int64) * 16)
torch.cuda.synchronize()
g = b[idx]                                    # 8-byte gathers (index kernel); writes 512 MiB contiguous
torch.cuda.synchronize()
print('stream read bytes', N * 4, 'stream write bytes', N // 2 * 8, 'gathers', M, 'gather useful bytes', M * 8,
      'index bytes read', M * 8, 'gather result bytes written', M * 8, float(s), int(g[0]))
