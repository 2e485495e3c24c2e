nproc
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, torch, time
sys.path.insert(0, '.')
import bench
from myriad_amd.synthetic import full_config
cfg = full_config()
for nt in (128, 64, 32, 16, 8):
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    r = bench.cpu_baseline("myriad", 1, cfg)
    print(nt, "threads:", r["value"], "img/s", "wall %.1f s" % (time.perf_counter() - t0), r["sample"][-70:], flush=True)
PY
