"""What a plain store stream reaches on this chip at the draw launch's size (38 MB written, 19 MB read) -- torch fill_ / copy_
as the yardstick.   usage: python tools/store_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
for mb in (4.8, 19.2, 38.3, 153, 613):
    n = int(mb * 1e6 / 2)
    t = torch.empty(n, dtype=torch.bfloat16, device=dev)
    us = bench._graph_time(lambda: t.fill_(1.0), dev)
    src = torch.empty(n // 2, dtype=torch.float32, device=dev)
    dst = torch.empty(n // 2, dtype=torch.float32, device=dev)
    us_c = bench._graph_time(lambda: dst.copy_(src), dev)
    print("%6.1f MB: fill_ %.2f us = %.2f TB/s written;  copy_ (same bytes read and written) %.2f us = %.2f TB/s moved"
          % (mb, us, mb / us, us_c, 2 * mb / us_c))
