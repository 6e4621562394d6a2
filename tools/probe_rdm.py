import sys
sys.path.insert(0, '/root/repo')
from quantum_computations_amd.device import DeviceState
n = 28
dev = DeviceState.random(n, 1)
for bits in ([0], [0, 1, 2], [0, 5, 12, 25], [0, 1, 2, 3, 4], [0, 1, 2, 3, 4, 5]):
    qs = [n - 1 - b for b in bits]
    for _ in range(3):
        dev.reduced_density(qs)
