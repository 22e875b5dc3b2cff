"""Clock stamps of the bit-flip sampler's 96 -> 32 tile convolution (a library built with -DIDM_TC_TIMELINE, given in ISINGDIFF_LIB):
median cycles between the stamps of wavefronts 0 and 3 over 64 mid-launch workgroups.
ISINGDIFF_LIB=.../lib_tc_tl.so python tools/experiments/tileconv_timeline.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffusion-model_amd"))
import numpy as np, torch
from isingdiff import bitflip
B = 8192
x0 = np.random.RandomState(0).randint(2, size=(B, 28, 28, 1)).astype(np.uint8)
model = bitflip.BitflipModel(bitflip.synthetic_weights(0), device=0, time_step=100, precision="bf16")
bitflip.diffusion_backward(x0, model, 100, return_trajectory=False)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["ISINGDIFF_LIB"])
buf = np.zeros((64, 2, 16), np.int64)
assert lib.idm_tc_timeline(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["start", "begin() done", "table + round 0 loads", "round 0 staged", "round 1 loads", "round 1 staged", "round 2 loads",
         "round 2 staged", "halo cleared", "barrier", "last piece: operands addressed", "last piece: last MFMA", "last piece: epilogue done",
         "run() done", "stores drained"]
for w in (0, 1):
    t = buf[:, w, :15].astype(np.float64)
    d = t - t[:, :1]
    print("wavefront", 3 * w, "median clock ticks since start (clock64), then the step")
    med = np.median(d, axis=0)
    for i, n in enumerate(names):
        print(f"  {i:2d} {n:36s} {med[i]:10.0f} {med[i] - (med[i - 1] if i else 0):10.0f}")
wall = buf[:, 0, 15]
print("wall_clock64 spread of the 64 starts:", int(wall.max() - wall.min()), "ticks of 10 ns")
