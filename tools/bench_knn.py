"""eval_knn.py consumer on one MI355X: features/s of extract_features (Swin-T backbone, 224^2, eval mode) and the k-NN scoring
rate (test rows/s against N_train stored features).  usage: bench_knn.py [N_train] [N_test] [C]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import esvit_amd
from esvit_amd import config as CFG
from esvit_amd import eval as E

dev = torch.device("cuda:0")
ntr = int(sys.argv[1]) if len(sys.argv) > 1 else 320000
nte = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
C = int(sys.argv[3]) if len(sys.argv) > 3 else 768
esvit_amd.set_precision("bf16")
model = esvit_amd.build_model(CFG.swin_config("swin_tiny_w7", DROP_PATH_RATE=0.0), is_teacher=True).to(dev).eval()
x = torch.randn(256, 3, 224, 224, device=dev)
with torch.no_grad():
    for _ in range(2):
        model(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        f = model(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
print("extract_features: %.0f images/s (Swin-T, 224^2, batch 256, bf16)" % (256 / dt))
g = torch.Generator(device=dev).manual_seed(0)
xtr = torch.nn.functional.normalize(torch.randn(ntr, C, device=dev, generator=g), dim=1)
xte = torch.nn.functional.normalize(torch.randn(nte, C, device=dev, generator=g), dim=1)
ytr = torch.randint(0, 1000, (ntr,), device=dev, generator=g)
yte = torch.randint(0, 1000, (nte,), device=dev, generator=g)
E.knn_classifier(xtr, ytr, xte[:200], yte[:200], 20, 0.07, num_chunks=2)
torch.cuda.synchronize()
t0 = time.time()
E.knn_classifier(xtr, ytr, xte, yte, 20, 0.07)
torch.cuda.synchronize()
dt = time.time() - t0
print("knn_classifier: %d test x %d train x %d: %.2f s, %.0f test rows/s, similarity %.1f TFLOP/s fp32-equivalent of wall" %
      (nte, ntr, C, dt, nte / dt, 2.0 * nte * ntr * C / dt / 1e12))
