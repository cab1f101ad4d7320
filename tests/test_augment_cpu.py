"""Crop producer, CPU side: the oracle (oracle/augment_ref.py) against Pillow -- live where Pillow is
importable, and through the committed fixtures tests/golden/augment_pil.npz (rendered by Pillow, oracle/gen_augment_golden.py) --;
the product's scalar arithmetic (esvit_amd/csrc/augment_math.h, compiled for the host) against the oracle; the product's
vectorised draw of the random parameters against the scalar restatement of torchvision's control flow.  All comparisons are
bit-exact."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import augment_ref as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "augment_pil.npz")

try:
    from PIL import Image, ImageEnhance, ImageFilter, ImageOps
except ImportError:  # pragma: no cover
    Image = None
needs_pil = pytest.mark.skipif(Image is None, reason="Pillow not importable")


def _img(rng, h, w, smooth=False):
    x = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if smooth and Image is not None:
        x = np.asarray(Image.fromarray(x).resize((w, h), Image.BICUBIC, box=(0, 0, w / 8, h / 8)))
    return np.ascontiguousarray(x)


# ---------------------------------------------------------------------------------------------------------------------
# oracle == Pillow fixtures (no Pillow needed)
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_matches_pillow_fixtures():
    g = np.load(GOLD)
    for k, row in enumerate(g["rows"]):
        final = g["final%d" % k]
        p = A.row_to_params(row, final.shape[0])
        assert (A.params_row(p, row[0]) == row).all()
        st = {}
        out = A.apply_crop(g["image%d" % row[0]], p, st)
        assert (st["color"] == g["color%d" % k]).all(), k
        assert (st["final_u8"] == final).all(), k
        assert out.dtype == np.float32 and out.shape == (3,) + final.shape[:2]
        r, ww, fw = A.box_weights(A.gaussian_box_radius(float(g["blur_radius%d" % k])))
        if p["blur"]:
            assert (r, ww, fw) == p["blur_box"]


# ---------------------------------------------------------------------------------------------------------------------
# oracle == Pillow, live
# ---------------------------------------------------------------------------------------------------------------------
@needs_pil
def test_oracle_resize_flip_matches_pillow():
    rng = np.random.default_rng(0)
    for (h, w, s) in [(375, 500, 224), (120, 90, 224), (300, 211, 96), (37, 41, 96), (224, 224, 224), (96, 500, 96), (700, 500, 96)]:
        img = _img(rng, h, w, smooth=h % 2 == 0)
        assert (A.resize_bicubic(img, s, s) == np.asarray(Image.fromarray(img).resize((s, s), Image.BICUBIC))).all(), (h, w, s)
    img = _img(rng, 375, 500)
    want = Image.fromarray(img).crop((57, 33, 57 + 150, 33 + 201)).resize((224, 224), Image.BICUBIC)
    assert (A.resized_crop(img, 33, 57, 201, 150, 224) == np.asarray(want)).all()
    assert (A.hflip(img) == np.asarray(Image.fromarray(img).transpose(Image.FLIP_LEFT_RIGHT))).all()


@needs_pil
def test_oracle_colour_ops_match_pillow():
    rng = np.random.default_rng(1)
    img = _img(rng, 120, 160, smooth=True)
    pim = Image.fromarray(img)
    assert (A.to_l(img) == np.asarray(pim.convert("L"))).all()
    for f in [0.6, 0.73312, 0.999, 1.0, 1.0001, 1.23456, 1.4, 0.0, 0.8, 1.2]:
        assert (A.adjust_brightness(img, f) == np.asarray(ImageEnhance.Brightness(pim).enhance(f))).all(), f
        assert (A.adjust_contrast(img, f) == np.asarray(ImageEnhance.Contrast(pim).enhance(f))).all(), f
        assert (A.adjust_saturation(img, f) == np.asarray(ImageEnhance.Color(pim).enhance(f))).all(), f
    d, i2 = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    dd, ii = np.repeat(d[..., None], 3, 2), np.repeat(i2[..., None], 3, 2)  # every (degenerate, image) byte pair
    for f in rng.uniform(0.5, 1.5, 6):
        assert (A.blend(dd, ii, float(f)) == np.asarray(Image.blend(Image.fromarray(dd), Image.fromarray(ii), float(f)))).all(), f
    # HSV both ways on a 1/5 sample of all colours (the generator script checks all 2^24)
    allc = np.arange(0, 1 << 24, 5, dtype=np.uint32)
    cols = np.stack([(allc >> 16) & 255, (allc >> 8) & 255, allc & 255], -1).astype(np.uint8)[None]
    assert (A.rgb_to_hsv(cols) == np.asarray(Image.fromarray(cols).convert("HSV"))).all()
    assert (A.hsv_to_rgb(cols) == np.asarray(Image.fromarray(cols, "HSV").convert("RGB"))).all()
    assert (A.to_grayscale3(img) == np.dstack([np.asarray(pim.convert("L"))] * 3)).all()
    assert (A.solarize(img) == np.asarray(ImageOps.solarize(pim))).all()


@needs_pil
def test_oracle_blur_matches_pillow():
    rng = np.random.default_rng(2)
    imgs = [_img(rng, 96, 96), _img(rng, 224, 224, smooth=True)]
    for r in list(rng.uniform(0.1, 2.0, 24)) + [0.1, 2.0, 1.0, 3.7]:
        for im in imgs:
            want = np.asarray(Image.fromarray(im).filter(ImageFilter.GaussianBlur(radius=float(r))))
            assert (A.gaussian_blur(im, float(r)) == want).all(), r


# ---------------------------------------------------------------------------------------------------------------------
# the product's arithmetic header, compiled for the host, == oracle
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def host_math(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("augmath") / "aug_math_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "esvit_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "aug_math_host.cpp"), "-o", so], check=True)
    return C.CDLL(so)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_header_resample_coefficients(host_math):
    for (n_in, n_out) in [(500, 224), (375, 224), (90, 224), (211, 96), (41, 96), (224, 224), (1000, 96), (17, 96), (23, 224), (2000, 224)]:
        bounds, kk = A.resample_coeffs(n_in, n_out)
        kmax = host_math.aug_t_ksize(n_in, n_out)
        assert kmax == kk.shape[1]
        b = np.zeros((n_out, 2), np.int32)
        k = np.zeros((n_out, kmax), np.int32)
        host_math.aug_t_coeffs(n_in, n_out, kmax, _ptr(b), _ptr(k))
        assert (b == bounds).all() and (k == kk).all(), (n_in, n_out)


def test_header_colour_arithmetic(host_math):
    rng = np.random.default_rng(3)
    d, i2 = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    d, i2 = np.ascontiguousarray(d), np.ascontiguousarray(i2)
    host_math.aug_t_blend.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_void_p]
    for f in list(rng.uniform(0.5, 1.5, 20)) + [0.0, 1.0, 0.6, 1.4]:
        out = np.empty_like(d)
        host_math.aug_t_blend(_ptr(d), _ptr(i2), d.size, float(np.float32(f)), _ptr(out))
        assert (out == A.blend(d, i2, float(f))).all(), f
    allc = np.arange(1 << 24, dtype=np.uint32)  # every colour
    cols = np.ascontiguousarray(np.stack([(allc >> 16) & 255, (allc >> 8) & 255, allc & 255], -1).astype(np.uint8))
    out = np.empty_like(cols)
    for fn, ref in [(host_math.aug_t_rgb_to_hsv, A.rgb_to_hsv), (host_math.aug_t_hsv_to_rgb, A.hsv_to_rgb)]:
        fn.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        fn(_ptr(cols), cols.shape[0], _ptr(out))
        for lo in range(0, 1 << 24, 1 << 22):
            assert (out[lo:lo + (1 << 22)] == ref(cols[None, lo:lo + (1 << 22)])[0]).all()
    lum = np.empty(cols.shape[0], np.uint8)
    host_math.aug_t_to_l.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    host_math.aug_t_to_l(_ptr(cols), cols.shape[0], _ptr(lum))
    assert (lum == A.to_l(cols)).all()


def test_header_blur_and_normalize(host_math):
    rng = np.random.default_rng(4)
    host_math.aug_t_box_line.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
    for radius in list(rng.uniform(0.1, 2.0, 10)) + [3.7, 6.0]:
        r, ww, fw = A.box_weights(A.gaussian_box_radius(radius))
        line = rng.integers(0, 256, (1, 131, 1), dtype=np.uint8)
        out = np.empty(131, np.uint8)
        host_math.aug_t_box_line(_ptr(line), 131, r, ww, fw, _ptr(out))
        assert (out == A.box_blur_axis1(line, r, ww, fw)[0, :, 0]).all(), radius
    v = np.arange(256, dtype=np.uint8)
    host_math.aug_t_normalize.argtypes = [C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_void_p]
    want = A.to_tensor_normalize(np.repeat(v[:, None, None], 3, 2))  # [3, 256, 1]
    for c in range(3):
        out = np.empty(256, np.float32)
        host_math.aug_t_normalize(_ptr(v), 256, A.MEAN[c], A.STD[c], _ptr(out))
        assert np.array_equal(out, want[c, :, 0])


# ---------------------------------------------------------------------------------------------------------------------
# the product's vectorised draws == torchvision's control flow, crop by crop
# ---------------------------------------------------------------------------------------------------------------------
def test_sampler_matches_scalar_restatement(lib_built):
    import esvit_amd.data as D
    rng = np.random.default_rng(5)
    n = 1500
    u = rng.random((n, D.NDRAWS))
    H, W = rng.integers(30, 600, n), rng.integers(30, 600, n)
    H[:100], W[:100] = 20, rng.integers(300, 900, 100)   # aspect ratios no attempt can satisfy: the central-crop fallback
    H[100:200], W[100:200] = rng.integers(300, 900, 100), 20
    fallbacks = 0
    for (scale, bp, sp, S) in [((0.4, 1.0), 1.0, 0.0, 224), ((0.4, 1.0), 0.1, 0.2, 224), ((0.05, 0.4), 0.5, 0.0, 96), ((0.9, 1.0), 0.5, 0.0, 96)]:
        rows = D.sample_params(u, np.arange(n), H, W, S, scale, bp, sp)
        assert rows.dtype == np.int32 and rows.shape == (n, 24)
        for i in range(n):
            p = A.sample_crop_params(u[i], int(H[i]), int(W[i]), S, scale, bp, sp)
            assert (A.params_row(p, src=i) == rows[i]).all(), (i, scale)
            fallbacks += (p["top"] == (H[i] - p["h"]) // 2 and p["left"] == (W[i] - p["w"]) // 2)
        assert (rows[:, 1] >= 0).all() and (rows[:, 1] + rows[:, 3] <= H).all() and (rows[:, 2] + rows[:, 4] <= W).all()
    assert fallbacks > 50
    # the marginal rates are the reference's (build.py:207-212,222,229-230,248)
    rows = D.sample_params(rng.random((20000, D.NDRAWS)), 0, np.full(20000, 375), np.full(20000, 500), 224, (0.4, 1.0), 0.1, 0.2)
    for col, want in [(5, 0.5), (14, 0.2), (18, 0.2)]:
        assert abs(rows[:, col].mean() - want) < 0.015
    assert abs((rows[:, 6] >= 0).mean() - 0.8) < 0.015 and abs((rows[:, 15] > 0).mean() - 0.1) < 0.01
    area = rows[:, 3] * rows[:, 4] / (375 * 500)
    assert 0.39 < area.min() and area.max() <= 1.0 and 0.55 < area.mean() < 0.7   # big boxes of a bad aspect ratio are re-drawn


def test_augmentation_object_layout(lib_built):
    import esvit_amd.data as D
    aug = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=0)
    assert [s[0] for s in aug.slots] == [224, 224] + [96] * 8 and aug.groups == {224: [0, 1], 96: list(range(2, 10))}
    assert [s[2:] for s in aug.slots[:3]] == [(1.0, 0.0), (0.1, 0.2), (0.5, 0.0)]
    aug2 = D.DataAugmentationDINO((0.14, 1.0), (0.05, 0.4), (4, 2), (96, 128))
    assert [s[0] for s in aug2.slots] == [224, 224, 96, 96, 96, 96, 128, 128]

    class Fake:
        H, W = np.array([300, 120, 64]), np.array([400, 90, 64])

        def __len__(self):
            return 3
    draws = aug.draw(Fake())
    assert draws[224][0].shape == (6, 24) and draws[96][0].shape == (24, 24)
    assert (draws[96][0][:, 0] == np.tile(np.arange(3), 8)).all()        # slot-major: crop slot c of image b at row c * B + b
    with pytest.raises(ValueError):
        D.PackedImages([np.zeros((4, 4), np.uint8)], device="cpu")


def test_collate_draws_in_the_worker(lib_built):
    """DataAugmentationDINO.collate: a picklable collate_fn that makes the draws from the image sizes (DataLoader worker side)"""
    import torch
    import esvit_amd.data as D
    aug = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=0)
    rng = np.random.default_rng(0)
    batch = [(rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8), i) for i, (h, w) in enumerate([(120, 160), (90, 64), (200, 200)])]
    (images, draws), labels = aug.collate(batch)
    assert len(images) == 3 and images[0].dtype == torch.uint8 and labels.tolist() == [0, 1, 2]
    assert set(draws) == {224, 96} and draws[224][0].shape == (6, 24) and draws[96][0].shape == (24, 24)
    for S, (rows, mh, mw) in draws.items():
        H, W = np.tile([120, 90, 200], len(rows) // 3), np.tile([160, 64, 200], len(rows) // 3)
        assert (rows[:, 1] + rows[:, 3] <= H).all() and (rows[:, 2] + rows[:, 4] <= W).all() and mh == rows[:, 3].max() and mw == rows[:, 4].max()
    loader = torch.utils.data.DataLoader(batch, batch_size=3, collate_fn=aug.collate, num_workers=1)  # crosses a process boundary
    (images2, draws2), _ = next(iter(loader))
    assert [tuple(i.shape) for i in images2] == [tuple(i.shape) for i in images] and set(draws2) == {224, 96}


@needs_pil
def test_oracle_full_pipeline_matches_pillow_random_cases():
    """whole crops -- random image sizes, boxes, operation orders, factors, blur radii -- rendered by the oracle and by the Pillow
    calls torchvision's PIL back end makes (oracle/gen_augment_golden.pil_crop): every uint8 stage identical"""
    from oracle.gen_augment_golden import pil_crop
    rng = np.random.default_rng(2025)
    for case in range(40):
        H, W = int(rng.integers(12, 260)), int(rng.integers(12, 260))
        img = _img(rng, H, W, smooth=bool(case % 2))
        S = int(rng.choice([32, 96, 224]) if case % 5 else rng.choice([20, 48, 100]))
        scale = (0.4, 1.0) if case % 3 else (0.05, 0.4)
        p = A.sample_crop_params(rng.random(36), H, W, S, scale, 0.6, 0.3)
        if case % 7 == 0:
            p["order"] = [int(i) for i in rng.permutation(4)]
        if case % 4 == 0 and (S > 24):
            p["blur"], p["blur_radius"] = True, float(rng.uniform(0.1, 4.0))
        st_pil, st_mine = {}, {}
        final = pil_crop(img, p, st_pil)
        A.apply_crop(img, p, st_mine)
        assert (st_mine["color"] == st_pil["color"]).all(), (case, p)
        assert (st_mine["final_u8"] == final).all(), (case, p)
        row = A.params_row(p, 0)                                   # and the parameter row round-trips to the same crop
        st_row = {}
        A.apply_crop(img, A.row_to_params(row, S), st_row)
        assert (st_row["final_u8"] == final).all(), case
