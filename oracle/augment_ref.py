"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the crop producer: a numpy restatement, in integer / float32
arithmetic, of what ``DataAugmentationDINO`` (datasets/build.py:203-261) does to ONE decoded RGB image given the random draws.

The reference composes third-party code that is not vendored in /root/reference:
  * torchvision.transforms (version unpinned by requirements.txt; the PIL back end ``functional_pil.py`` of the 0.8-0.15
    series): ``RandomResizedCrop`` = ``img.crop(box).resize((S, S), BICUBIC)``, ``RandomHorizontalFlip`` = FLIP_LEFT_RIGHT,
    ``ColorJitter`` = ``ImageEnhance.Brightness / Contrast / Color`` and the HSV hue rotation in a random order,
    ``RandomGrayscale`` = ``convert("L")`` replicated, ``ToTensor`` = u8 / 255, ``Normalize`` = (x - mean) / std;
  * Pillow (``Image.resize``: libImaging/Resample.c 8 bpc two-pass convolution with 22-bit fixed-point coefficients;
    ``Image.blend``: Blend.c; ``convert``: Convert.c rgb2l / rgb2hsv / hsv2rgb; ``ImageFilter.GaussianBlur``: BoxBlur.c,
    three box passes per axis with 24-bit fixed-point weights; ``ImageOps.solarize``), called from utils.py:43-75.
Pillow IS importable in the build image (12.2.0): tests/test_augment_cpu.py pins every function below against Pillow itself --
live, and through the committed fixtures tests/golden/augment_pil.npz written by oracle/gen_augment_golden.py -- bit for bit
(the HSV round trip over all 2^24 colours).  torchvision is absent: the composition and the parameter draws follow its published
algorithm (cited per function); their parity is therefore anchored on Pillow for the arithmetic and on the reference's call site
(datasets/build.py:206-246) for the order, probabilities and ranges.

Images are HWC uint8 numpy arrays.  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this module.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c
MEAN = (0.485, 0.456, 0.406)  # datasets/build.py:215
STD = (0.229, 0.224, 0.225)


# ---------------------------------------------------------------------------------------------
# Image.resize(size, BICUBIC) -- Resample.c: precompute_coeffs + normalize_coeffs_8bpc + the two passes
# ---------------------------------------------------------------------------------------------
def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """bounds [out, 2] (first tap, number of taps) and the fixed-point taps [out, ksize] of one axis (box = the whole axis)"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, out_size):
    """convolve along axis 0 of a [n, ...] uint8 array"""
    bounds, kk = resample_coeffs(img.shape[0], out_size)
    src = img.astype(np.int64)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = np.tensordot(kk[xx, :n], src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def resize_bicubic(img, out_w, out_h):
    """``Image.resize((out_w, out_h), BICUBIC)`` of an HWC uint8 image: horizontal pass (uint8 result), then vertical"""
    tmp = _resample_axis0(np.ascontiguousarray(img.transpose(1, 0, 2)), out_w).transpose(1, 0, 2)
    return _resample_axis0(np.ascontiguousarray(tmp), out_h)


def resized_crop(img, top, left, h, w, size):
    """torchvision functional_pil: ``crop`` (= ``img.crop((left, top, left + w, top + h))``) then ``resize`` -- the filter sees
    the crop only, its taps are clamped at the crop's border (datasets/build.py:220,227,246)"""
    return resize_bicubic(img[top:top + h, left:left + w], size, size)


def hflip(img):
    """RandomHorizontalFlip -> ``img.transpose(FLIP_LEFT_RIGHT)`` (datasets/build.py:207)"""
    return img[:, ::-1].copy()


# ---------------------------------------------------------------------------------------------
# ColorJitter / RandomGrayscale -- ImageEnhance over Image.blend, Convert.c
# ---------------------------------------------------------------------------------------------
def to_l(img):
    """Convert.c rgb2l: (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16"""
    v = img.astype(np.int64)
    return ((v[..., 0] * 19595 + v[..., 1] * 38470 + v[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """``Image.blend(deg, img, alpha)`` (Blend.c): float32 ``in1 + alpha * (in2 - in1)``, truncated; clipped only when
    extrapolating (alpha outside [0, 1]); alpha 0 / 1 return copies (Image.py)"""
    if alpha == 0.0:
        return deg.copy()
    if alpha == 1.0:
        return img.copy()
    a = np.float32(alpha)
    d = img.astype(np.int32) - deg.astype(np.int32)
    t = deg.astype(np.float32) + a * d.astype(np.float32)  # float32 product, float32 sum
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(img, f):
    """ImageEnhance.Brightness: degenerate = black"""
    return blend(np.zeros_like(img), img, f)


def adjust_contrast(img, f):
    """ImageEnhance.Contrast: degenerate = grey of ``int(mean(L) + 0.5)`` (ImageStat mean = sum / count in double)"""
    lum = to_l(img)
    mean = int(int(lum.astype(np.int64).sum()) / lum.size + 0.5)
    return blend(np.full_like(img, mean), img, f)


def adjust_saturation(img, f):
    """ImageEnhance.Color: degenerate = ``convert("L").convert("RGB")``"""
    return blend(np.repeat(to_l(img)[..., None], 3, axis=2), img, f)


def rgb_to_hsv(img):
    """Convert.c rgb2hsv_row (float32 intermediates, truncating casts)"""
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(np.float32)
    rc = (maxc - r).astype(np.float32) / safe
    gc = (maxc - g).astype(np.float32) / safe
    bc = (maxc - b).astype(np.float32) / safe
    rc64, gc64, bc64 = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)  # "2.0 + rc - bc" is a double expression
    h = np.where(r == maxc, (bc - gc).astype(np.float64), np.where(g == maxc, 2.0 + rc64 - bc64, 4.0 + gc64 - rc64)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)  # double expression stored to a float
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    grey = maxc == minc
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], axis=-1).astype(np.uint8)


def hsv_to_rgb(hsv):
    """Convert.c hsv2rgb (float32 h, f, fs; double products; C ``round``)"""
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    vf = v.astype(np.float32).astype(np.float64)
    f64, fs64 = f.astype(np.float64), fs.astype(np.float64)

    def c_round(x):  # C round(): half away from zero; the arguments are >= 0
        return np.floor(x + 0.5).astype(np.int32)
    p = np.clip(c_round(vf * (1.0 - fs64)), 0, 255)
    q = np.clip(c_round(vf * (1.0 - fs64 * f64)), 0, 255)
    t = np.clip(c_round(vf * (1.0 - fs64 * (1.0 - f64))), 0, 255)
    sel = i % 6
    r = np.choose(sel, [v, q, p, p, t, v])
    g = np.choose(sel, [t, v, v, q, p, p])
    b = np.choose(sel, [p, p, t, v, v, q])
    grey = s == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], axis=-1).astype(np.uint8)


def hue_delta(hue_factor):
    """functional_pil.adjust_hue: ``np.uint8(hue_factor * 255)`` -- a C cast (truncation toward zero), modulo 256"""
    return int(hue_factor * 255) & 255


def adjust_hue(img, hue_factor):
    """functional_pil.adjust_hue: RGB -> HSV, h += uint8(hue_factor * 255) (wrapping), HSV -> RGB"""
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_delta(hue_factor)).astype(np.uint8)
    return hsv_to_rgb(hsv)


def to_grayscale3(img):
    """RandomGrayscale -> ``to_grayscale(img, 3)``: L replicated (datasets/build.py:212)"""
    return np.repeat(to_l(img)[..., None], 3, axis=2)


# ---------------------------------------------------------------------------------------------
# utils.GaussianBlur (utils.py:43-61) -> ImageFilter.GaussianBlur -> BoxBlur.c; utils.Solarization (utils.py:64-75)
# ---------------------------------------------------------------------------------------------
def gaussian_box_radius(radius, passes=3):
    """BoxBlur.c _gaussian_blur_radius: float variables, double expressions"""
    f32 = np.float32
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32((2 * l + 1) * (l * (l + 1) - 3 * sigma2))
    a = f32(a / f32(6 * (sigma2 - (l + 1) * (l + 1))))
    return f32(l + a)


def box_weights(float_radius):
    """ImagingHorizontalBoxBlur: integer radius, the weight of a full tap and of the two fractional far taps (24-bit fixed point)"""
    fr = np.float32(float_radius)
    radius = int(fr)
    ww = int(np.float32(1 << 24) / (fr * np.float32(2) + np.float32(1)))  # UINT32 / float -> float, truncated
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def box_blur_axis1(img, radius, ww, fw):
    """one ImagingLineBoxBlur pass along axis 1: full taps x-r..x+r and the two far taps, indices clamped to the line"""
    n = img.shape[1]
    assert n > radius + 1
    src = img.astype(np.int64)
    x = np.arange(n)
    acc = np.zeros_like(src)
    for d in range(-radius, radius + 1):
        acc += src[:, np.clip(x + d, 0, n - 1)]
    far = src[:, np.clip(x - radius - 1, 0, n - 1)] + src[:, np.clip(x + radius + 1, 0, n - 1)]
    return ((acc * ww + far * fw + (1 << 23)) >> 24).astype(np.uint8)


def gaussian_blur(img, radius, passes=3, box=None):
    """``img.filter(ImageFilter.GaussianBlur(radius))``: three horizontal box passes, then three vertical ones (uint8 between
    passes); ``box`` = (r, ww, fw) replaces the radius when the weights are already known (a parameter row)"""
    r, ww, fw = box if box is not None else box_weights(gaussian_box_radius(radius, passes))
    out = img
    for _ in range(passes):
        out = box_blur_axis1(out, r, ww, fw)
    out = out.transpose(1, 0, 2)
    for _ in range(passes):
        out = box_blur_axis1(out, r, ww, fw)
    return np.ascontiguousarray(out.transpose(1, 0, 2))


def solarize(img, threshold=128):
    """ImageOps.solarize: values >= threshold are inverted"""
    return np.where(img < threshold, img, 255 - img).astype(np.uint8)


def to_tensor_normalize(img):
    """ToTensor (u8 -> float32 / 255, CHW) then Normalize ((x - mean) / std in float32) (datasets/build.py:213-216)"""
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    mean = np.asarray(MEAN, np.float32)[:, None, None]
    std = np.asarray(STD, np.float32)[:, None, None]
    return (x - mean) / std


# ---------------------------------------------------------------------------------------------
# one crop, given its draws (the dict produced by sample_crop_params / esvit_amd.data)
# ---------------------------------------------------------------------------------------------
def apply_crop(img, p, stages=None):
    """datasets/build.py:219-250: RandomResizedCrop -> flip -> ColorJitter (random order) -> grayscale -> blur -> solarize ->
    ToTensor + Normalize.  ``stages``, when a dict, receives the uint8 image after the resize / colour / blur stages."""
    x = resized_crop(img, p["top"], p["left"], p["h"], p["w"], p["size"])
    if p["flip"]:
        x = hflip(x)
    if stages is not None:
        stages["resize"] = x.copy()
    for op in p["order"]:  # empty when RandomApply(p=0.8) skipped the jitter
        if op == 0:
            x = adjust_brightness(x, p["brightness"])
        elif op == 1:
            x = adjust_contrast(x, p["contrast"])
        elif op == 2:
            x = adjust_saturation(x, p["saturation"])
        elif op == 3:
            x = adjust_hue(x, p["hue"])
    if p["gray"]:
        x = to_grayscale3(x)
    if stages is not None:
        stages["color"] = x.copy()
    if p["blur"]:
        x = gaussian_blur(x, p.get("blur_radius"), box=p.get("blur_box"))
    if p["solarize"]:
        x = solarize(x)
    if stages is not None:
        stages["final_u8"] = x.copy()
    return to_tensor_normalize(x)


# ---------------------------------------------------------------------------------------------
# the random draws of one crop, literally as torchvision / utils.py make them, from a row of uniforms
# ---------------------------------------------------------------------------------------------
def sample_crop_params(u, H, W, size, scale, blur_p, solarize_p, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """One crop's draws from ``u`` (36 uniforms in [0, 1), the layout of esvit_amd/data.py), following the control flow of
    torchvision's RandomResizedCrop.get_params (10 attempts, central-crop fallback), RandomHorizontalFlip (``rand < 0.5``),
    RandomApply (skip when ``p < rand``), ColorJitter.get_params (randperm + four uniform factors), RandomGrayscale
    (``rand < 0.2``) and utils.py:52-59,72 (``random() <= p``, ``uniform(0.1, 2.)``, ``random() < p``).  Returns the dict
    apply_crop consumes.  uniform_(a, b) = a + (b - a) * u; randint(0, n) = floor(u * n); randperm = argsort of 4 uniforms."""
    u = [float(x) for x in u]
    area = H * W
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    box = None
    for k in range(10):
        target_area = area * (scale[0] + (scale[1] - scale[0]) * u[2 * k])
        aspect_ratio = math.exp(log_ratio[0] + (log_ratio[1] - log_ratio[0]) * u[2 * k + 1])
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= W and 0 < h <= H:
            box = (int(math.floor(u[20] * (H - h + 1))), int(math.floor(u[21] * (W - w + 1))), h, w)
            break
    if box is None:
        in_ratio = float(W) / float(H)
        if in_ratio < min(ratio):
            w = W
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = H
            w = int(round(h * max(ratio)))
        else:
            w, h = W, H
        box = ((H - h) // 2, (W - w) // 2, h, w)
    p = {"top": box[0], "left": box[1], "h": box[2], "w": box[3], "size": size, "flip": u[22] < 0.5}
    f32 = np.float32

    def uniform(a, b, x):  # a float32 draw, as float(torch.empty(1).uniform_(a, b))
        return float(f32(a) + f32(b - a) * f32(x))
    p["order"] = [int(i) for i in np.argsort(np.asarray(u[24:28]), kind="stable")] if not (0.8 < u[23]) else []
    p["brightness"], p["contrast"] = uniform(0.6, 1.4, u[28]), uniform(0.6, 1.4, u[29])
    p["saturation"], p["hue"] = uniform(0.8, 1.2, u[30]), uniform(-0.1, 0.1, u[31])
    p["gray"] = u[32] < 0.2
    p["blur"] = u[33] <= blur_p
    p["blur_radius"] = 0.1 + (2.0 - 0.1) * u[34]
    p["solarize"] = u[35] < solarize_p
    return p


def params_row(p, src=0):
    """the int32 row of include/esvit_hip.h (esvit_aug_crops) for the draws ``p``"""
    row = np.zeros(24, np.int32)
    row[0:6] = (src, p["top"], p["left"], p["h"], p["w"], int(p["flip"]))
    row[6:10] = (list(p["order"]) + [-1] * 4)[:4]
    row[10:13] = np.asarray([p["brightness"], p["contrast"], p["saturation"]], np.float32).view(np.int32)
    row[13] = hue_delta(p["hue"])
    row[14] = int(p["gray"])
    if p["blur"]:
        r, ww, fw = p["blur_box"] if "blur_box" in p else box_weights(gaussian_box_radius(p["blur_radius"]))
        row[15:18] = (r + 1, ww, fw)
    row[18] = int(p["solarize"])
    return row


def row_to_params(row, size):
    """the draws of a parameter row (the inverse of params_row; the blur comes back as its box weights)"""
    row = np.asarray(row, np.int32)
    f = row[10:13].view(np.float32)
    hue = int(row[13])
    p = {"top": int(row[1]), "left": int(row[2]), "h": int(row[3]), "w": int(row[4]), "size": size, "flip": bool(row[5]),
         "order": [int(o) for o in row[6:10] if o >= 0], "brightness": float(f[0]), "contrast": float(f[1]), "saturation": float(f[2]),
         "hue": (hue if hue < 128 else hue - 256) / 255.0 + (1e-9 if hue < 128 else -1e-9), "gray": bool(row[14]), "blur": bool(row[15] > 0),
         "blur_box": (int(row[15]) - 1, int(row[16]), int(row[17])), "solarize": bool(row[18])}
    assert hue_delta(p["hue"]) == hue
    return p
