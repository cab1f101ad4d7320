"""Crop producer on the GPU: the reference's ``DataAugmentationDINO`` (datasets/build.py:203-261) for a whole batch of decoded
images resident in HBM, emitting the multi-crop list ``[B,3,224,224] x 2 + [B,3,96,96] x 8`` the training step consumes
(main_esvit.py:541-547) directly in device memory.

The reference transforms ONE PIL image per call inside 10 DataLoader workers (main_esvit.py:198); here the dataset's transform
only decodes (uint8 HWC), the random draws of all ``B x ncrops`` crops are made on the host in one vectorised pass
(:func:`sample_params`: the algorithms of torchvision's ``RandomResizedCrop.get_params``, ``RandomHorizontalFlip``, ``RandomApply``,
``ColorJitter.get_params``, ``RandomGrayscale`` and of utils.py:43-75, same probabilities and ranges) and ``esvit_aug_crops``
(csrc/augment.hip) renders them, bit-exact with Pillow's arithmetic for the same draws.

There is no CPU path: the arithmetic lives in the HIP library only.
"""
import math

import numpy as np
import torch

from . import ops

NDRAWS = 36  # uniforms consumed per crop, layout below
U_ATTEMPT, U_I, U_J, U_FLIP, U_APPLY, U_PERM, U_BRIGHT, U_CONTRAST, U_SAT, U_HUE, U_GRAY, U_BLUR_P, U_BLUR_R, U_SOL = (
    0, 20, 21, 22, 23, 24, 28, 29, 30, 31, 32, 33, 34, 35)
RATIO = (3.0 / 4.0, 4.0 / 3.0)  # RandomResizedCrop default
BRIGHTNESS, CONTRAST, SATURATION, HUE = (0.6, 1.4), (0.6, 1.4), (0.8, 1.2), (-0.1, 0.1)  # ColorJitter(0.4, 0.4, 0.2, 0.1), build.py:209
P_FLIP, P_JITTER, P_GRAY = 0.5, 0.8, 0.2  # build.py:207-212
BLUR_RADIUS = (0.1, 2.0)  # utils.py:47


def box_blur_weights(radius, passes=3):
    """Pillow's BoxBlur.c for ``ImageFilter.GaussianBlur(radius)``: the box radius of each of the three passes (float variables,
    double expressions) and ImagingHorizontalBoxBlur's integer radius / 24-bit weights -> (r, ww, fw) of esvit_aug_crops"""
    f32 = np.float32
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32((2 * l + 1) * (l * (l + 1) - 3 * sigma2))
    a = f32(a / f32(6 * (sigma2 - (l + 1) * (l + 1))))
    fr = f32(l + a)
    r = int(fr)
    ww = int(f32(1 << 24) / (fr * f32(2) + f32(1)))
    fw = ((1 << 24) - (2 * r + 1) * ww) // 2
    return r, ww, fw


def sample_params(u, src, H, W, size, scale, blur_p, solarize_p):
    """int32 [n, 24] parameter rows of esvit_aug_crops from uniforms ``u`` [n, NDRAWS] in [0, 1): crop n is cut from image
    ``src[n]`` of ``H[n] x W[n]`` pixels.  ``blur_p`` / ``solarize_p``: probabilities of the slot (build.py:222,229-230,248)."""
    u = np.asarray(u, np.float64)
    n = u.shape[0]
    H, W = np.asarray(H, np.int64), np.asarray(W, np.int64)
    rows = np.zeros((n, ops.AUG_PARAM_INTS), np.int32)
    rows[:, 0] = src
    # RandomResizedCrop.get_params: up to 10 draws of (area, log-uniform aspect); the first box that fits wins
    area = (H * W).astype(np.float64)[:, None]
    ua, ur = u[:, U_ATTEMPT:U_ATTEMPT + 20:2], u[:, U_ATTEMPT + 1:U_ATTEMPT + 20:2]
    target = area * (scale[0] + (scale[1] - scale[0]) * ua)
    lr0, lr1 = math.log(RATIO[0]), math.log(RATIO[1])
    aspect = np.exp(lr0 + (lr1 - lr0) * ur)
    w = np.rint(np.sqrt(target * aspect)).astype(np.int64)  # round(): half to even, as Python's
    h = np.rint(np.sqrt(target / aspect)).astype(np.int64)
    ok = (w > 0) & (w <= W[:, None]) & (h > 0) & (h <= H[:, None])
    first = np.argmax(ok, axis=1)
    found = ok[np.arange(n), first]
    cw, ch = w[np.arange(n), first], h[np.arange(n), first]
    top = np.floor(u[:, U_I] * (H - ch + 1)).astype(np.int64)  # torch.randint(0, H - h + 1)
    left = np.floor(u[:, U_J] * (W - cw + 1)).astype(np.int64)
    # fallback: the central crop of the closest admissible aspect ratio
    in_ratio = W / H
    fw_ = np.where(in_ratio > RATIO[1], np.rint(H * RATIO[1]).astype(np.int64), W)
    fh_ = np.where(in_ratio < RATIO[0], np.rint(W / RATIO[0]).astype(np.int64), H)
    cw, ch = np.where(found, cw, fw_), np.where(found, ch, fh_)
    top, left = np.where(found, top, (H - ch) // 2), np.where(found, left, (W - cw) // 2)
    rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4] = top, left, ch, cw
    rows[:, 5] = u[:, U_FLIP] < P_FLIP
    # RandomApply(ColorJitter, p = 0.8): a random order of the four operations, one factor each
    apply = u[:, U_APPLY] <= P_JITTER
    order = np.argsort(u[:, U_PERM:U_PERM + 4], axis=1, kind="stable")
    rows[:, 6:10] = np.where(apply[:, None], order, -1)

    def factor(col, rng):
        return (np.float32(rng[0]) + np.float32(rng[1] - rng[0]) * u[:, col].astype(np.float32)).astype(np.float32)
    rows[:, 10] = factor(U_BRIGHT, BRIGHTNESS).view(np.int32)
    rows[:, 11] = factor(U_CONTRAST, CONTRAST).view(np.int32)
    rows[:, 12] = factor(U_SAT, SATURATION).view(np.int32)
    hue = factor(U_HUE, HUE).astype(np.float64)
    rows[:, 13] = np.trunc(hue * 255).astype(np.int64) & 255  # functional_pil.adjust_hue: np.uint8(hue_factor * 255)
    rows[:, 14] = u[:, U_GRAY] < P_GRAY
    # utils.GaussianBlur: random() <= p, radius uniform(0.1, 2.0); utils.Solarization: random() < p
    blur = u[:, U_BLUR_P] <= blur_p
    radius = BLUR_RADIUS[0] + (BLUR_RADIUS[1] - BLUR_RADIUS[0]) * u[:, U_BLUR_R]
    for i in np.nonzero(blur)[0]:
        r, ww, fw = box_blur_weights(radius[i])
        rows[i, 15], rows[i, 16], rows[i, 17] = r + 1, ww, fw
    rows[:, 18] = u[:, U_SOL] < solarize_p
    return rows


class PackedImages:
    """decoded RGB images of one batch in device memory: ``data`` uint8 (HWC, back to back), ``table`` int64 [B, 3] =
    (byte offset, H, W) -- the input format of esvit_aug_crops"""

    def __init__(self, images, device="cuda"):
        arrs, H, W = [], [], []
        for im in images:
            if isinstance(im, torch.Tensor):
                t = im
            else:  # PIL.Image / numpy HWC
                t = torch.from_numpy(np.ascontiguousarray(np.asarray(im)))
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("PackedImages: expected uint8 H x W x 3 (decoded RGB) images, got %s %s" % (t.dtype, tuple(t.shape)))
            arrs.append(t.reshape(-1))
            H.append(int(t.shape[0]))
            W.append(int(t.shape[1]))
        self.H, self.W = np.asarray(H, np.int64), np.asarray(W, np.int64)
        sizes = self.H * self.W * 3
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        arrs.append(torch.zeros(4, dtype=torch.uint8, device=arrs[0].device))  # the kernels read whole dwords: pad to the boundary
        on_dev = all(a.is_cuda for a in arrs)
        if on_dev:
            self.data = torch.cat(arrs)
        else:
            host = torch.cat([a.cpu() for a in arrs])
            self.data = host.pin_memory().to(device, non_blocking=True) if torch.cuda.is_available() else host.to(device)
        self.table = torch.from_numpy(np.stack([offs, self.H, self.W], axis=1).astype(np.int64)).to(self.data.device)

    def __len__(self):
        return len(self.H)


class DataAugmentationDINO:
    """``DataAugmentationDINO(global_crops_scale, local_crops_scale, local_crops_number, local_crops_size)`` (datasets/
    build.py:203-250), called on a BATCH: ``aug(images)`` takes B decoded images (a :class:`PackedImages` or a list of uint8 HWC
    tensors / arrays / PIL images) and returns the reference's collated crop list -- 2 + sum(local_crops_number) float32 CUDA
    tensors ``[B, 3, S, S]``, crop slot c of image b at ``out[c][b]``."""

    def __init__(self, global_crops_scale, local_crops_scale, local_crops_number, local_crops_size=(96,), seed=None, device="cuda"):
        local_crops_number = [local_crops_number] if isinstance(local_crops_number, int) else list(local_crops_number)
        local_crops_size = [local_crops_size] if isinstance(local_crops_size, int) else list(local_crops_size)
        if len(local_crops_number) != len(local_crops_size):
            raise ValueError("local_crops_number and local_crops_size must have one entry per local resolution")
        self.device = device
        self.rng = np.random.default_rng(seed)
        # (output size, scale range, blur probability, solarize probability) per crop slot, in the reference's order
        self.slots = [(224, tuple(global_crops_scale), 1.0, 0.0),   # build.py:219-224
                      (224, tuple(global_crops_scale), 0.1, 0.2)]   # build.py:226-232
        for n_crop, size in zip(local_crops_number, local_crops_size):  # build.py:244-250, 257-260
            self.slots += [(int(size), tuple(local_crops_scale), 0.5, 0.0)] * int(n_crop)
        self.groups = {}  # output size -> slots, rendered by one esvit_aug_crops call
        for c, s in enumerate(self.slots):
            self.groups.setdefault(s[0], []).append(c)

    def draw(self, packed, uniforms=None):
        """the parameter rows of every size group: {S: (int32 [len(slots) * B, 24], max_h, max_w)} (slot-major)"""
        B = len(packed)
        out = {}
        for S, slots in self.groups.items():
            rows = []
            for c in slots:
                _, scale, blur_p, sol_p = self.slots[c]
                u = self.rng.random((B, NDRAWS)) if uniforms is None else uniforms[c]
                rows.append(sample_params(u, np.arange(B), packed.H, packed.W, S, scale, blur_p, sol_p))
            rows = np.concatenate(rows)
            out[S] = (rows, int(rows[:, 3].max()), int(rows[:, 4].max()))
        return out

    def __call__(self, images, uniforms=None):
        packed = images if isinstance(images, PackedImages) else PackedImages(images, self.device)
        B = len(packed)
        crops = [None] * len(self.slots)
        for S, (rows, max_h, max_w) in self.draw(packed, uniforms).items():
            params = torch.from_numpy(rows).pin_memory().to(packed.data.device, non_blocking=True)
            nbytes = rows.shape[0] * (3 * S * S + 4)
            planes = ops.workspace((nbytes + 3) // 4, packed.data.device, slot="aug_planes").view(torch.uint8)  # stream-ordered scratch
            out, _ = ops.aug_crops(packed.data, packed.table, params, S, max_h, max_w, planes=planes)
            for k, c in enumerate(self.groups[S]):
                crops[c] = out[k * B:(k + 1) * B]
        return crops


class GpuAugmentedLoader:
    """wraps a DataLoader whose dataset only DECODES (``(list of uint8 HWC images, labels)`` per batch, e.g. with
    ``collate_fn=lambda b: ([x for x, _ in b], torch.tensor([y for _, y in b]))``) into the iterator ``train_one_epoch`` expects
    (main_esvit.py:522): ``(crops, labels)`` with the crops produced on the GPU"""

    def __init__(self, loader, augment):
        self.loader, self.augment = loader, augment

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for images, labels in self.loader:
            yield self.augment(images), labels
