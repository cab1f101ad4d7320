"""Crop producer on the MI355X (esvit_aug_crops through esvit_amd.ops / esvit_amd.data) against Pillow's own output (fixtures)
and against the CPU oracle -- every stage BIT-EXACT (tolerance 0: uint8 planes and the float32 crops) --, plus size-independent
properties at the benchmark batch (B = 128, 2 x 224^2 + 8 x 96^2 crops per image)."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "augment_pil.npz")


def _render(images, rows, S):
    """rows through the HIP path: (out [n, 3, S, S] float32, planes [n, 3, S, S] uint8) as numpy"""
    from esvit_amd import data as D, ops
    packed = D.PackedImages([torch.from_numpy(im) for im in images])
    params = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).cuda()
    out, planes = ops.aug_crops(packed.data, packed.table, params, S, int(rows[:, 3].max()), int(rows[:, 4].max()))
    torch.cuda.synchronize()
    return out.cpu().numpy(), planes.cpu().numpy()


def _check_against_oracle(images, rows, S, out, planes):
    for k, row in enumerate(rows):
        st = {}
        want = A.apply_crop(images[row[0]], A.row_to_params(row, S), st)
        got_planes = planes[k].transpose(1, 2, 0)
        assert (got_planes == st["resize"]).all(), ("resize stage", k, int((got_planes != st["resize"]).sum()))
        assert np.array_equal(out[k], want), ("final", k, float(np.abs(out[k] - want).max()))


def test_crops_match_pillow_fixtures(lib_built):
    g = np.load(GOLD)
    images = [g["image%d" % i] for i in range(int(g["n_images"]))]
    rows = g["rows"]
    sizes = np.array([g["final%d" % k].shape[0] for k in range(len(rows))])
    for S in sorted(set(sizes)):
        idx = np.nonzero(sizes == S)[0]
        out, _ = _render(images, rows[idx], int(S))
        unfinished = rows[idx].copy()
        unfinished[:, 15], unfinished[:, 18] = 0, 0   # the same draws without blur / solarize: the crop after the jitter
        mid, _ = _render(images, unfinished, int(S))
        for j, k in enumerate(idx):
            assert np.array_equal(mid[j], A.to_tensor_normalize(g["color%d" % k])), ("jitter stage vs Pillow", k)
            assert np.array_equal(out[j], A.to_tensor_normalize(g["final%d" % k])), ("final vs Pillow", k)


def test_crops_match_oracle_random_batch(lib_built):
    from esvit_amd import data as D
    rng = np.random.default_rng(11)
    shapes = [(375, 500), (500, 333), (64, 64), (120, 90), (33, 200), (224, 224)]
    images = [np.ascontiguousarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in shapes]
    for im in images[:3]:  # smooth content in some: saturated / grey areas exercise the HSV branches
        im[:] = (np.add.outer(np.arange(im.shape[0]) * 3, np.arange(im.shape[1]) * 2)[..., None] // np.array([3, 5, 7]) % 256).astype(np.uint8)
    aug = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=3)
    packed = D.PackedImages([torch.from_numpy(im) for im in images])
    draws = aug.draw(packed)
    crops = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=3)(packed)   # same seed: same draws
    assert len(crops) == 10 and [tuple(c.shape) for c in crops] == [(6, 3, 224, 224)] * 2 + [(6, 3, 96, 96)] * 8
    assert all(c.is_cuda and c.dtype == torch.float32 for c in crops)
    for S, (rows, _, _) in draws.items():
        out, planes = _render(images, rows, S)
        _check_against_oracle(images, rows, S, out, planes)
        for k, c in enumerate(aug.groups[S]):  # the object returns the same crops, slot-major
            assert np.array_equal(crops[c].cpu().numpy(), out[k * 6:(k + 1) * 6])


def test_every_blur_radius_path(lib_built):
    """box radii 0..4 run in the tiled kernel (one template instance each), larger ones in the whole-plane kernel"""
    rng = np.random.default_rng(14)
    img = np.ascontiguousarray(rng.integers(0, 256, (300, 260, 3), dtype=np.uint8))
    img[60:200, 40:180] = (np.add.outer(np.arange(140), np.arange(140))[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8)
    rows, radii = [], [0.1, 0.7, 1.1, 2.0, 2.6, 3.4, 4.1, 5.2, 6.0, 9.0]
    for k, radius in enumerate(radii):
        p = A.sample_crop_params(rng.random(36), 300, 260, 224, (0.4, 1.0), 1.0, 0.3)
        p.update(blur=True, blur_radius=radius, solarize=bool(k % 2))
        rows.append(A.params_row(p, 0))
    rows = np.stack(rows)
    assert sorted(set(rows[:, 15] - 1)) == [0, 1, 2, 3, 4, 5, 8], sorted(set(rows[:, 15] - 1))
    for S in (224, 96):
        out, planes = _render([img], rows, S)
        _check_against_oracle([img], rows, S, out, planes)


def test_large_boxes_use_smaller_tiles(lib_built):
    """a 1500 x 1100 box resized to 96 (scale 15.6) does not fit the 32 x 32 tile's LDS: the 16 / 8 tiles must agree with the oracle"""
    from esvit_amd import ops
    rng = np.random.default_rng(12)
    img = np.ascontiguousarray(rng.integers(0, 256, (1500, 1100, 3), dtype=np.uint8))
    p = A.sample_crop_params(rng.random(36), 1500, 1100, 96, (0.99, 1.0), 0.0, 0.0)
    p.update(top=0, left=0, h=1500, w=1100, order=[], gray=False, blur=False)
    p2 = dict(p, h=700, w=420, top=100, left=50, flip=True)
    rows = np.stack([A.params_row(p, 0), A.params_row(p2, 0)])
    out, planes = _render([img], rows, 96)
    _check_against_oracle([img], rows, 96, out, planes)
    assert ops.aug_max_box(96) >= 1500 and ops.aug_max_box(224) >= 2000
    # beyond the staged tiles (scale 29: the source rows of even an 8 x 8 tile do not fit the LDS): the unstaged 8 x 8 variant
    big = np.ascontiguousarray(rng.integers(0, 256, (2800, 2400, 3), dtype=np.uint8))
    pb = dict(p, top=10, left=20, h=2780, w=2300, flip=False)
    rows_b = np.stack([A.params_row(pb, 0)])
    out_b, planes_b = _render([big], rows_b, 96)
    _check_against_oracle([big], rows_b, 96, out_b, planes_b)
    with pytest.raises(RuntimeError):
        _render([img], np.stack([A.params_row(dict(p, h=1 << 20), 0)]), 96)


def test_properties_at_benchmark_batch(lib_built):
    """B = 128, the crop list of BASELINE.json's step: (i) an S x S image cropped whole with no operation drawn is ToTensor +
    Normalize of the source, exactly; (ii) flipping is an involution on the rendered crop; (iii) two renderings of the same draws
    are identical; (iv) every value is a normalised byte"""
    from esvit_amd import data as D, ops
    rng = np.random.default_rng(13)
    B = 128
    images = [torch.from_numpy(rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8)).cuda()
              for h, w in zip(rng.integers(200, 520, B), rng.integers(200, 520, B))]
    aug = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=5)
    packed = D.PackedImages(images)
    draws = aug.draw(packed)
    for S, (rows, mh, mw) in draws.items():
        params = torch.from_numpy(rows).cuda()
        out, _ = ops.aug_crops(packed.data, packed.table, params, S, mh, mw)
        out2, _ = ops.aug_crops(packed.data, packed.table, params, S, mh, mw)
        assert torch.equal(out, out2)                                                                   # (iii)
        flipped = rows.copy()
        flipped[:, 5] ^= 1
        plain = flipped.copy()
        plain[:, 6:10], plain[:, 14], plain[:, 15], plain[:, 18] = -1, 0, 0, 0                          # geometry only
        a, _ = ops.aug_crops(packed.data, packed.table, torch.from_numpy(plain).cuda(), S, mh, mw)
        plain[:, 5] ^= 1
        b, _ = ops.aug_crops(packed.data, packed.table, torch.from_numpy(plain).cuda(), S, mh, mw)
        assert torch.equal(a, b.flip(-1))                                                               # (ii)
        lut = torch.from_numpy(np.stack([A.to_tensor_normalize(np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 3, 2))[c, :, 0]
                                         for c in range(3)])).cuda()                                    # [3, 256]
        for c in range(3):                                                                              # (iv)
            assert torch.isin(out[:, c].reshape(-1)[::97], lut[c]).all()
    # (i)
    S = 224
    src = [torch.from_numpy(rng.integers(0, 256, (S, S, 3), dtype=np.uint8)).cuda() for _ in range(B)]
    packed = D.PackedImages(src)
    rows = np.zeros((B, 24), np.int32)
    rows[:, 0], rows[:, 3], rows[:, 4], rows[:, 6:10] = np.arange(B), S, S, -1
    out, planes = ops.aug_crops(packed.data, packed.table, torch.from_numpy(rows).cuda(), S, S, S)
    hwc = torch.stack(src)
    assert torch.equal(planes, hwc.permute(0, 3, 1, 2))
    for c in range(3):  # ToTensor + Normalize of every byte value, from the oracle (torch's own GPU division by 255 is a multiplication)
        assert torch.equal(out[:, c], lut[c][hwc[..., c].long()])


def test_rejects_host_tensors(lib_built):
    from esvit_amd import ops
    with pytest.raises((AssertionError, RuntimeError)):
        ops.aug_crops(torch.zeros(12, dtype=torch.uint8), torch.zeros((1, 3), dtype=torch.int64), torch.zeros((1, 24), dtype=torch.int32), 96, 2, 2)


def test_prefetching_loader_equals_inline_rendering(lib_built):
    """GpuAugmentedLoader renders batch n + 1 on its own stream while batch n is consumed: same crops as rendering inline"""
    from esvit_amd import data as D
    rng = np.random.default_rng(21)
    batches = []
    for _ in range(4):
        imgs = [torch.from_numpy(rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8)) for h, w in zip(rng.integers(60, 200, 5), rng.integers(60, 200, 5))]
        batches.append((imgs, torch.arange(5)))

    def run(prefetch):
        aug = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=9)
        out = []
        for crops, labels in D.GpuAugmentedLoader(batches, aug, prefetch=prefetch):
            acc = torch.zeros((), device="cuda")
            for c in crops:                       # consume on the current stream, as a training step would
                acc = acc + c.double().sum()
            out.append(([c.clone() for c in crops], labels, acc))
        torch.cuda.synchronize()
        return out
    a, b = run(True), run(False)
    assert len(a) == len(b) == 4
    for (ca, la, sa), (cb, lb, sb) in zip(a, b):
        assert torch.equal(la, lb) and float(sa) == float(sb)
        assert all(torch.equal(x, y) for x, y in zip(ca, cb))


def test_single_image_call_matches_the_reference_contract(lib_built):
    """aug(image) -- one PIL image / array, as a Dataset transform would call it -- returns 2 + 8 tensors [3, S, S]: the rows of the
    batched call with the same draws"""
    from esvit_amd import data as D
    rng = np.random.default_rng(31)
    img = np.ascontiguousarray(rng.integers(0, 256, (180, 240, 3), dtype=np.uint8))
    one = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=2)(img)
    batch = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=2)([img])
    assert len(one) == 10 and [tuple(c.shape) for c in one] == [(3, 224, 224)] * 2 + [(3, 96, 96)] * 8
    assert all(torch.equal(a, b[0]) for a, b in zip(one, batch))
    try:
        from PIL import Image
    except ImportError:
        return
    pil = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=2)(Image.fromarray(img))
    assert all(torch.equal(a, b) for a, b in zip(one, pil))
