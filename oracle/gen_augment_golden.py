"""Writes tests/golden/augment_pil.npz: crops rendered by PILLOW ITSELF (the arithmetic the reference's DataAugmentationDINO runs,
datasets/build.py:203-261) for fixed draws -- the fixtures that pin oracle/augment_ref.py and, through it, the HIP crop producer.

    python oracle/gen_augment_golden.py

The composition below is torchvision's PIL back end written out with the Pillow calls it makes (functional_pil.py: crop, resize,
hflip, adjust_brightness / contrast / saturation / hue, to_grayscale) plus utils.py:43-75; torchvision itself is not installed.
Before writing, the script also checks the oracle against Pillow exhaustively where that is possible (all 2^24 colours through
the HSV round trip, all 2^16 (degenerate, image) byte pairs through Image.blend)."""
import os
import sys

import numpy as np
from PIL import Image, ImageEnhance, ImageFilter, ImageOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import augment_ref as A  # noqa: E402


def pil_hue(img, hue_factor):  # functional_pil.adjust_hue
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.array(hue_factor * 255).astype("uint8")
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def pil_crop(img, p, stages=None):
    """one crop of DataAugmentationDINO with Pillow, for the draws ``p`` (the dict of augment_ref.sample_crop_params)"""
    x = Image.fromarray(img).crop((p["left"], p["top"], p["left"] + p["w"], p["top"] + p["h"]))
    x = x.resize((p["size"], p["size"]), Image.BICUBIC)
    if p["flip"]:
        x = x.transpose(Image.FLIP_LEFT_RIGHT)
    for op in p["order"]:
        if op == 0:
            x = ImageEnhance.Brightness(x).enhance(p["brightness"])
        elif op == 1:
            x = ImageEnhance.Contrast(x).enhance(p["contrast"])
        elif op == 2:
            x = ImageEnhance.Color(x).enhance(p["saturation"])
        else:
            x = pil_hue(x, p["hue"])
    if p["gray"]:
        x = Image.fromarray(np.dstack([np.array(x.convert("L"))] * 3))
    if stages is not None:
        stages["color"] = np.asarray(x).copy()
    if p["blur"]:
        x = x.filter(ImageFilter.GaussianBlur(radius=p["blur_radius"]))
    if p["solarize"]:
        x = ImageOps.solarize(x)
    return np.asarray(x).copy()


def source_images():
    rng = np.random.default_rng(2024)
    out = []
    for (h, w, smooth) in [(150, 200, 6), (97, 64, 1), (260, 180, 12), (40, 300, 3)]:
        x = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if smooth > 1:  # low-frequency content (a blown-up corner) + a little noise: edges, flat areas and saturated colours
            x = np.asarray(Image.fromarray(x).resize((w, h), Image.BICUBIC, box=(0, 0, w / smooth, h / smooth))).astype(np.int16)
            x = np.clip(x + rng.integers(-6, 7, x.shape), 0, 255).astype(np.uint8)
        out.append(np.ascontiguousarray(x))
    return out


def cases(images):
    """(image index, draws) -- sampled from fixed uniforms, then edited so that every branch is hit at least once"""
    rng = np.random.default_rng(7)
    out = []
    spec = [(0, 224, (0.4, 1.0), 1.0, 0.0), (2, 224, (0.4, 1.0), 0.1, 0.2), (1, 224, (0.4, 1.0), 1.0, 0.0)]
    spec += [(i % 4, 96, (0.05, 0.4), 0.5, 0.0) for i in range(7)]
    for k, (src, S, scale, bp, sp) in enumerate(spec):
        H, W = images[src].shape[:2]
        p = A.sample_crop_params(rng.random(36), H, W, S, scale, bp, sp)
        if k == 1:
            p.update(solarize=True, blur=True, blur_radius=0.1, order=[3, 1, 0, 2], gray=False)
        if k == 2:
            p.update(order=[1, 2, 3, 0], flip=True, blur_radius=2.0)
        if k == 3:
            p.update(order=[], gray=False, blur=False, flip=False)          # the resize alone
        if k == 4:
            p.update(order=[2, 0, 3, 1], gray=True, blur=True, blur_radius=1.37)
        if k == 5:
            p.update(order=[0, 1, 2, 3], brightness=1.4, contrast=0.6, saturation=1.2, hue=-0.1, blur=False)
        if k == 6:
            p.update(top=3, left=5, h=17, w=23, order=[3], hue=0.1, blur=True, blur_radius=0.61)   # a box smaller than the output
        if k == 9:
            p = A.sample_crop_params(rng.random(36), 40, 300, 96, (0.9, 1.0), 0.5, 0.0)            # central-crop fallback
        out.append((src, p))
    return out


def exhaustive_checks():
    allc = np.arange(1 << 24, dtype=np.uint32)
    cols = np.stack([(allc >> 16) & 255, (allc >> 8) & 255, allc & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    assert (A.rgb_to_hsv(cols) == np.asarray(Image.fromarray(cols).convert("HSV"))).all()
    assert (A.hsv_to_rgb(cols) == np.asarray(Image.fromarray(cols, "HSV").convert("RGB"))).all()
    d, i2 = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    dd, ii = np.repeat(d[..., None], 3, 2), np.repeat(i2[..., None], 3, 2)
    for f in np.random.default_rng(0).uniform(0.5, 1.5, 16):
        assert (A.blend(dd, ii, float(f)) == np.asarray(Image.blend(Image.fromarray(dd), Image.fromarray(ii), float(f)))).all()
    print("exhaustive HSV / blend checks: oracle == Pillow")


def main():
    exhaustive_checks()
    images = source_images()
    store = {"n_images": np.int64(len(images))}
    for i, im in enumerate(images):
        store["image%d" % i] = im
    rows = []
    for k, (src, p) in enumerate(cases(images)):
        stages = {}
        final = pil_crop(images[src], p, stages)
        mine_stages = {}
        mine = A.apply_crop(images[src], p, mine_stages)
        assert (mine_stages["final_u8"] == final).all() and (mine_stages["color"] == stages["color"]).all(), k
        assert np.array_equal(mine, A.to_tensor_normalize(final))
        rows.append(A.params_row(p, src))
        store["color%d" % k], store["final%d" % k] = stages["color"], final
        store["blur_radius%d" % k] = np.float64(p["blur_radius"])
    store["rows"] = np.stack(rows)
    path = os.path.join(ROOT, "tests", "golden", "augment_pil.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(rows), "crops")


if __name__ == "__main__":
    main()
