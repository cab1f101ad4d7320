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


def synthetic_crops(B, n_local=8, seed=1234, sizes=(224, 96)):
    """torch.randn multi-crop batch in the reference's list order [global, global, local x n_local] (main_esvit.py:513: ten
    tensors, 2 x (B,3,224,224) + 8 x (B,3,96,96)): the synthetic input of bench.py, of smoke() and of the parity tests."""
    g = torch.Generator().manual_seed(seed)
    crops = [torch.randn(B, 3, sizes[0], sizes[0], generator=g) for _ in range(2)]
    crops += [torch.randn(B, 3, sizes[1], sizes[1], generator=g) for _ in range(n_local)]
    return crops


def box_blur_weights(radius, passes=3):
    """Pillow's BoxBlur.c for ``ImageFilter.GaussianBlur(radius)``: the box radius of each of the three passes (float variables,
    double expressions) and ImagingHorizontalBoxBlur's integer radius / 24-bit weights -> (r, ww, fw) of esvit_aug_crops;
    ``radius`` may be an array"""
    f32, f64 = np.float32, np.float64
    radius = np.asarray(radius, f64).astype(f32)
    sigma2 = (radius * radius / f32(passes)).astype(f32)
    L = np.sqrt(12.0 * sigma2.astype(f64) + 1.0).astype(f32)
    l = np.floor((L.astype(f64) - 1.0) / 2.0).astype(f32)
    a = ((2 * l + 1) * (l * (l + 1) - 3 * sigma2)).astype(f32)
    a = (a / (6 * (sigma2 - (l + 1) * (l + 1))).astype(f32)).astype(f32)
    fr = (l + a).astype(f32)
    r = fr.astype(np.int64)
    ww = (f32(1 << 24) / (fr * f32(2) + f32(1))).astype(f32).astype(np.int64)
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
    s0, s1 = np.reshape(np.asarray(scale[0], np.float64), (-1, 1)), np.reshape(np.asarray(scale[1], np.float64), (-1, 1))  # scalar or per crop
    target = area * (s0 + (s1 - s0) * ua)
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
    r, ww, fw = box_blur_weights(radius)
    rows[:, 15], rows[:, 16], rows[:, 17] = np.where(blur, r + 1, 0), np.where(blur, ww, 0), np.where(blur, fw, 0)
    rows[:, 18] = u[:, U_SOL] < solarize_p
    return rows


class PackedImages:
    """decoded RGB images of one batch in device memory: ``data`` uint8 (HWC, back to back), ``table`` int64 [B, 3] =
    (byte offset, H, W) -- the input format of esvit_aug_crops"""

    def __init__(self, images, device="cuda"):
        if len(images) == 0:
            raise ValueError("PackedImages: an empty batch")
        arrs, H, W = [], [], []
        for im in images:
            if isinstance(im, torch.Tensor):
                t = im
            else:  # PIL.Image / numpy HWC
                t = torch.from_numpy(np.array(im))  # (a copy: arrays viewed from PIL images are read-only)
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
    tensors ``[B, 3, S, S]``, crop slot c of image b at ``out[c][b]``.  Called on ONE image it returns the per-sample list of
    ``[3, S, S]`` tensors, as the reference's transform does."""

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
            k = len(slots)
            u = self.rng.random((k * B, NDRAWS)) if uniforms is None else np.concatenate([uniforms[c] for c in slots])
            per_slot = lambda i: np.repeat(np.asarray([self.slots[c][i] for c in slots], np.float64), B, axis=0)  # noqa: E731
            scale = per_slot(1)
            rows = sample_params(u, np.tile(np.arange(B), k), np.tile(packed.H, k), np.tile(packed.W, k), S, (scale[:, 0], scale[:, 1]),
                                 per_slot(2), per_slot(3))  # one vectorised pass over the slots of a size
            out[S] = (rows, int(rows[:, 3].max()), int(rows[:, 4].max()))
        return out

    def collate(self, batch):
        """``collate_fn`` for a DataLoader whose dataset yields ``(uint8 HWC image, label)``: the DataLoader WORKER makes the random
        draws (they need the image sizes only), the training process just uploads and renders -> ``((images, draws), labels)``"""
        images = [torch.as_tensor(np.asarray(x)) for x, _ in batch]

        class _Sizes:
            H, W = np.asarray([im.shape[0] for im in images]), np.asarray([im.shape[1] for im in images])

            def __len__(self):
                return len(images)
        return (images, self.draw(_Sizes())), torch.as_tensor([y for _, y in batch])

    def __call__(self, images, uniforms=None, draws=None):
        if not isinstance(images, (list, tuple, PackedImages)):
            # one decoded image (PIL.Image / uint8 H x W x 3 array or tensor), the reference's per-sample contract (build.py:252-261):
            # the crop list without the batch axis
            return [c[0] for c in self.__call__([images], uniforms=uniforms, draws=draws)]
        packed = images if isinstance(images, PackedImages) else PackedImages(images, self.device)
        B = len(packed)
        dev = packed.data.device
        if draws is None:
            draws = self.draw(packed, uniforms)
        # one host-to-device copy of all parameter rows, out of a pinned staging buffer that is reused every other call
        total = sum(rows.shape[0] for rows, _, _ in draws.values())
        stage, ready = self._staging(total, dev)
        ready.synchronize()  # the copy issued two calls ago has long finished
        at = 0
        for rows, _, _ in draws.values():
            stage[at:at + rows.shape[0]] = torch.from_numpy(rows)
            at += rows.shape[0]
        params = stage[:total].to(dev, non_blocking=True)
        ready.record()
        crops = [None] * len(self.slots)
        at = 0
        for S, (rows, max_h, max_w) in draws.items():
            n = rows.shape[0]
            planes = ops.workspace((n * (3 * S * S + 4) + 3) // 4, dev, slot="aug_planes").view(torch.uint8)  # stream-ordered scratch
            out, _ = ops.aug_crops(packed.data, packed.table, params[at:at + n], S, max_h, max_w, planes=planes)
            at += n
            for k, c in enumerate(self.groups[S]):
                crops[c] = out[k * B:(k + 1) * B]
        return crops

    def _staging(self, rows, dev):
        if dev.type != "cuda":
            raise RuntimeError("esvit_amd.data: the crop producer runs on the GPU only")
        bufs = self.__dict__.setdefault("_pinned", [])
        if not bufs or bufs[0][0].shape[0] < rows:
            bufs[:] = [(torch.empty((rows, ops.AUG_PARAM_INTS), dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
        self._turn = 1 - getattr(self, "_turn", 0)
        return bufs[self._turn]


class GpuAugmentedLoader:
    """wraps a DataLoader whose dataset only DECODES (``collate_fn=augment.collate``: the workers also make the random draws; or any
    collate that yields ``(list of uint8 HWC images, labels)``) into the iterator ``train_one_epoch`` expects (main_esvit.py:522):
    ``(crops, labels)`` with the crops produced on the GPU.  With ``prefetch`` (default) batch n + 1 is uploaded and rendered on a
    second HIP stream while the training step of batch n runs (the producer's kernels are VALU-bound byte arithmetic, the step's are
    memory / MFMA work: they share the chip well), and the consumer's stream only waits for an event."""

    def __init__(self, loader, augment, prefetch=True):
        self.loader, self.augment, self.prefetch = loader, augment, prefetch
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def _render(self, item):
        images, labels = item
        if isinstance(images, tuple) and len(images) == 2 and isinstance(images[1], dict):  # DataAugmentationDINO.collate
            return self.augment(images[0], draws=images[1]), labels
        return self.augment(images), labels

    def __iter__(self):
        if not self.prefetch:
            for item in self.loader:
                yield self._render(item)
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        it = iter(self.loader)

        def produce():
            try:
                item = next(it)
            except StopIteration:
                return None
            self._stream.wait_stream(torch.cuda.current_stream())  # (a scratch freed by the consumer is not reused too early)
            with torch.cuda.stream(self._stream):
                crops, labels = self._render(item)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            return crops, labels, ev
        nxt = produce()
        while nxt is not None:
            crops, labels, ev = nxt
            nxt = produce()  # enqueued BEFORE the consumer's step: it runs beside it
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            for c in crops:
                c.record_stream(main)
            yield crops, labels
