"""development aid: per-phase s_memtime totals of bb_sweep_kernel (library built with -DPL_SWEEP_TIMING=1)"""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, ".")
from pylinac_amd import _lib, winston_lutz
from pylinac_amd.synthetic import wl_frames
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
fr = torch.from_numpy(wl_frames(256, noise_sigma=noise)).cuda()
winston_lutz.analyze_batch(fr, 1 / 0.336, 5.0)
torch.cuda.synchronize()
lib = ctypes.CDLL(str(_lib.lib_path()))
out = (ctypes.c_ulonglong * 8)()
assert lib.pl_debug_sweep_timing(out) == 0
v = np.array(list(out), dtype=np.float64) / 256
names = ["runs/merge/table/screen", "crop mask", "flood fill", "holes/perimeter/moments", "hull (one lane)", "inside hull", "predicates+rest", "loop head"]
tot = v.sum()
print(f"noise {noise}: ticks per frame {tot:.0f}")
for n, x in zip(names, v):
    print(f"  {n:28s} {x:10.0f}  {100 * x / tot:5.1f} %")
