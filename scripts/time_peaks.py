import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from pylinac_amd import ops
from pylinac_amd.pipeline import EpidPipeline
from pylinac_amd.synthetic import epid_open_field_frames
dev = torch.device("cuda:0")
fr = epid_open_field_frames(256, 1024, 1024, device=dev)
p = EpidPipeline(256, 1024, 1024, dev); res = p.run(fr); prof = res.profile.clone()
def t(f, reps=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
r = ops.find_peaks_batch(prof, fwxm_height=0.5); print("peaks per profile (no filter):", r.count.float().mean().item())
print("fwxm top1 (height -inf)      us:", t(lambda: ops.find_peaks_batch(prof, cap=1, fwxm_height=0.5, max_number=1)))
print("all peaks, no top-k          us:", t(lambda: ops.find_peaks_batch(prof, cap=600, fwxm_height=0.5)))
print("threshold 0.5 top1           us:", t(lambda: ops.find_peaks_batch(prof, cap=1, threshold=0.5, fwxm_height=0.5, max_number=1)))
print("threshold 0.999 top1         us:", t(lambda: ops.find_peaks_batch(prof, cap=1, threshold=0.999, fwxm_height=0.5, max_number=1)))
sm = torch.from_numpy(np.tile(np.sin(np.linspace(0, 20, 1024)) + 2, (256, 1))).to(dev)
print("smooth sine (3 peaks)        us:", t(lambda: ops.find_peaks_batch(sm, cap=1, fwxm_height=0.5, max_number=1)))
