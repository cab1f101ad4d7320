import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_gemm as bg
from esvit_amd import ops
shapes = [s for s in bg.step_shapes(128) if (s[0] in ("s2 fc2", "s2 fc1", "s2 proj", "s3 fc2", "s3 fc1") and s[1] in ("fwd", "dgrad"))]
for name, kind, M, N, K, fl in shapes:
    r = bg.run_shape(name, kind, M, N, K, fl, (0,))
    print(name, kind, fl, round(r[0][0] * 1e6, 1))
