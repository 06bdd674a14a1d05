#!/usr/bin/env python3
"""Known-byte-count reads for calibrating rocprofv3's FETCH_SIZE on the access widths the engine uses (run under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE`): 2, 4 and 16 bytes per lane over a 2-GiB buffer, three launches each."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

L = ctypes.CDLL(os.path.join(ROOT, "evogp_amd", "lib", "libevogp_hip.so"))
buf = torch.randint(0, 255, (2 << 30,), dtype=torch.uint8, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for w in (2, 4, 16):
    for _ in range(3):
        rc = L.evogp_hip_debug_calibrate_read(ctypes.c_void_p(buf.data_ptr()), ctypes.c_ulonglong(buf.numel()), w, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(s))
        assert rc == 0
torch.cuda.synchronize()
print("bytes per launch", buf.numel())
