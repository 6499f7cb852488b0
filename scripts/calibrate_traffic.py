"""Calibration stream for rocprofv3's FETCH_SIZE / WRITE_SIZE in OUR access pattern (4 B per lane, coalesced
256-byte rows - the pattern of every state read/write of the step kernel): n floats read + n floats written
per dispatch by the library's elementwise test kernel.  MI355X_MICROARCH.md (HBM section): the counters are
calibrated only for 16 B/lane streaming reads; other widths must be calibrated on a known byte count."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd import _abi
lib = _abi.load_library()
lib.vmas_debug_math.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
n = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = torch.rand(n, device="cuda:0")
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(reps):
    assert lib.vmas_debug_math(0, x.data_ptr(), None, y.data_ptr(), n, None) == 0
torch.cuda.synchronize()
print("bytes_read_per_dispatch", 4 * n, "bytes_written_per_dispatch", 4 * n)
