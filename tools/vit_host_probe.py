"""wall time per step vs host enqueue time per step of the deit_tiny training step, ragged vs per-group (is a step bound by the GPU
or by the launching thread?).   python tools/vit_host_probe.py [--arch deit_tiny]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="deit_tiny")
    ap.add_argument("--batch", type=int, default=128)
    args = ap.parse_args()
    import esvit_amd
    from esvit_amd.engine import EsvitTrainer
    from tests import golden_utils as GU
    esvit_amd.set_precision("bf16")
    dev = torch.device("cuda:0")
    crops = [c.to(dev) for c in GU.make_crops(args.batch, seed=1)]
    for ragged in (True, False):
        torch.manual_seed(0)
        student, teacher, loss_fn = bench.build(dev, 0.1, args.arch)
        student.ragged_multi_crop = teacher.ragged_multi_crop = ragged
        tr = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1)
        for _ in range(4):
            tr.step(crops, 5e-4, 0.04, 0.996, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step(crops, 5e-4, 0.04, 0.996, 1)
        host = (time.perf_counter() - t0) / 10
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 10
        per = []
        for _ in range(24):  # per-step wall times with a synchronisation after every step: does the step time drift?
            t1 = time.perf_counter()
            tr.step(crops, 5e-4, 0.04, 0.996, 1)
            torch.cuda.synchronize()
            per.append(round((time.perf_counter() - t1) * 1e3, 1))
        print(json.dumps({"arch": args.arch, "ragged": ragged, "wall_ms": wall * 1e3, "host_enqueue_ms": host * 1e3, "per_step_synced_ms": per,
                          "mem_GB": torch.cuda.max_memory_allocated() / 1e9}))
        del tr, student, teacher, loss_fn
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
