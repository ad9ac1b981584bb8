#!/usr/bin/env python3
"""The scattered-8-byte-load probe alone (aoclgpu_probe_scatter8 over a 2 GiB buffer): run it under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace` to calibrate that counter for the codecs' candidate-window pattern
(tools/round_profile.sh does; tools/collect_profiles.py turns the result into profiles/traffic.json "scatter8_calibration")."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aocl_compression_amd as a
eng = a.engine(0)
n = 2 << 30
buf = torch.ones(n, dtype=torch.uint8, device=eng.device)
tot = C.c_uint64(0)
ms = eng.lib.aoclgpu_probe_scatter8(buf.data_ptr(), n, 256, C.byref(tot), eng._stream())
print(json.dumps({"loads": tot.value, "ms": ms, "buffer_bytes": n}))
