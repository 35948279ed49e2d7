import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle as O
from onepiece_amd import synthetic as S
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "OMP env", {k: v for k, v in os.environ.items() if k.startswith("OMP") or k.startswith("GOMP")})
fr = [(S.room_pose(i),) + S.room_render(S.room_pose(i)) for i in range(12)]
for th in (1, 8, 16, 32, 64, 128):
    O.set_fusion_threads(th)
    ov = O.Volume(voxel_res=0.005)
    t = time.perf_counter()
    for pose, d, c in fr: ov.integrate(d, c, pose)
    print(th, "threads: %.1f ms/frame" % ((time.perf_counter() - t) / len(fr) * 1e3))
