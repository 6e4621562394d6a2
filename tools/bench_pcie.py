import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from quantum_computations_amd.device import DeviceState
dev = DeviceState.random(28, 1)
out = {}
for trial in range(3):
    t0 = time.perf_counter(); host = dev.to_numpy(); t_down = time.perf_counter() - t0
    t0 = time.perf_counter(); dev.upload(host); t_up = time.perf_counter() - t0
    t0 = time.perf_counter(); again = dev.to_numpy(); t_down2 = time.perf_counter() - t0
    assert np.array_equal(host, again)
    print(json.dumps({"trial": trial, "download_fresh_buffer_GBps": host.nbytes / t_down / 1e9, "upload_GBps": host.nbytes / t_up / 1e9,
                      "download_s": t_down, "upload_s": t_up}))
    del host, again
