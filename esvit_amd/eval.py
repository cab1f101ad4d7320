"""Inference consumers of an EsViT checkpoint: the feature extraction and the weighted k-NN classifier
of the reference's eval_knn.py, on the HIP path.

``extract_features`` (eval_knn.py:165-190) runs the backbone (``build_model(config, is_teacher=True)`` with
``NUM_CLASSES 0``: the pooled cls features) batch by batch and gathers features and dataset indices of all ranks into
rank 0's feature matrix.  ``knn_classifier`` (eval_knn.py:193-232) scores every test feature against ALL train features
(cosine similarity of L2-normalised rows), keeps the k nearest, and lets them vote with weight exp(similarity / T).  The
similarity product -- the only heavy step: 50k x 1.28M x C for ImageNet -- is this library's fp32 MFMA GEMM reading the
train matrix in its stored [N_train, C] layout (no transposed copy); top-k, the one-hot vote and the ranking are
PyTorch indexing ops with the reference's tie behaviour.
"""
import os

import torch
import torch.distributed as dist

from . import ops


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def load_pretrained_weights(model, pretrained_weights, checkpoint_key, model_name=None, patch_size=None):
    """utils.load_pretrained_weights (utils.py:78-103): take `checkpoint_key` ("teacher" / "student") out of a training checkpoint
    written by main_esvit.py (utils.save_on_master), strip the DistributedDataParallel prefix, load non-strictly (the heads of the
    checkpoint have no place in a NUM_CLASSES 0 backbone).  Without a file the reference tries to download DINO's public ViT
    weights; there is no network on this side, so that branch leaves the random initialisation and says so.  Returns the
    load_state_dict message (the reference prints it)."""
    if not os.path.isfile(pretrained_weights):
        print("no checkpoint file at %r: the backbone keeps its random initialisation (the reference would fetch DINO's public weights "
              "here; this side has no network)" % (pretrained_weights,))
        return None
    ckpt = torch.load(pretrained_weights, map_location="cpu", weights_only=False)
    weights = ckpt.get(checkpoint_key, ckpt) if (checkpoint_key is not None and isinstance(ckpt, dict)) else ckpt
    if weights is not ckpt:
        print("checkpoint %s: using the %r entry" % (pretrained_weights, checkpoint_key))
    ddp_prefix = "module."  # utils.save_on_master stores the DistributedDataParallel-wrapped student
    weights = {(name[len(ddp_prefix):] if name.startswith(ddp_prefix) else name).replace("." + ddp_prefix, "."): tensor for name, tensor in weights.items()}
    report = model.load_state_dict(weights, strict=False)
    print("checkpoint %s loaded (non-strict): %s" % (pretrained_weights, report))
    return report


@torch.no_grad()
def extract_features(model, data_loader, use_cuda=True):
    """data_loader yields (samples, index) with ``index`` the position in ``data_loader.dataset`` (the reference's
    ReturnIndexDataset, eval_knn.py:235-238).  Returns the [len(dataset), C] feature matrix on rank 0, None elsewhere."""
    rank = dist.get_rank() if _dist_on() else 0
    world = dist.get_world_size() if _dist_on() else 1
    features = None
    dev = next(model.parameters()).device  # the reference hard-wires .cuda(); the model's device is the same thing on a GPU box
    for samples, index in data_loader:
        samples = samples.to(dev, non_blocking=True)
        index = index.to(dev, non_blocking=True)
        feats = model(samples).clone()
        if rank == 0 and features is None:
            features = torch.zeros(len(data_loader.dataset), feats.shape[-1])
            if use_cuda:
                features = features.to(dev, non_blocking=True)
        if world > 1:
            idx_l = [torch.empty_like(index) for _ in range(world)]
            dist.all_gather(idx_l, index)
            feat_l = [torch.empty_like(feats) for _ in range(world)]
            dist.all_gather(feat_l, feats)
            index_all, feats_all = torch.cat(idx_l), torch.cat(feat_l)
        else:
            index_all, feats_all = index, feats
        if rank == 0:
            if use_cuda:
                features.index_copy_(0, index_all, feats_all.float())
            else:
                features.index_copy_(0, index_all.cpu(), feats_all.float().cpu())
    return features


@torch.no_grad()
def knn_classifier(train_features, train_labels, test_features, test_labels, k, T, num_classes=1000, num_chunks=100):
    """-> (top1 %, top5 %).  Features are L2-normalised fp32 rows ([N_train, C], [N_test, C]) on the GPU.  The test set is
    scored in ``num_chunks`` chunks exactly as the reference does (eval_knn.py:197-198)."""
    train_features = train_features.float().contiguous()
    test_features = test_features.float().contiguous()
    train_labels, test_labels = train_labels.to(train_features.device), test_labels.to(train_features.device)
    top1, top5, total = 0.0, 0.0, 0
    num_test = test_labels.shape[0]
    per_chunk = max(1, num_test // num_chunks)  # (the reference divides by 100 unguarded and fails below 100 test images)
    for idx in range(0, num_test, per_chunk):
        features = test_features[idx:min(idx + per_chunk, num_test)]
        targets = test_labels[idx:min(idx + per_chunk, num_test)]
        bs = targets.shape[0]
        similarity = ops.linear_fwd(features, train_features)  # [bs, N_train] = features @ train^T, fp32 MFMA GEMM
        distances, indices = similarity.topk(k, largest=True, sorted=True)
        neighbors = torch.gather(train_labels.view(1, -1).expand(bs, -1), 1, indices)
        one_hot = torch.zeros(bs * k, num_classes, device=similarity.device)
        one_hot.scatter_(1, neighbors.view(-1, 1), 1)
        weights = distances.clone().div_(T).exp_()
        probs = torch.sum(one_hot.view(bs, -1, num_classes) * weights.view(bs, -1, 1), 1)
        _, predictions = probs.sort(1, True)
        correct = predictions.eq(targets.view(-1, 1))
        top1 += correct.narrow(1, 0, 1).sum().item()
        top5 += correct.narrow(1, 0, min(5, num_classes)).sum().item()
        total += bs
    return top1 * 100.0 / total, top5 * 100.0 / total


# ------------------------------------------------------------------------------------------------------------------------
# eval_linear.py: the linear probe on frozen features (eval_linear.py:244-325)
# ------------------------------------------------------------------------------------------------------------------------
class _ProbeLinearFn(torch.autograd.Function):
    """y = x W^T + b on the library's fp32 MFMA GEMM (forward, data gradient is not needed -- the features are frozen --,
    weight gradient with the bias gradient fused)"""

    @staticmethod
    def forward(ctx, x, W, b):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.linear_fwd(x, W.detach().contiguous(), b.detach())

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        dW, db = ops.linear_wgrad(gy.contiguous(), x, want_bias=True)
        return None, dW, db


class LinearClassifier(torch.nn.Module):
    """eval_linear.py:307-321: one nn.Linear on the flattened frozen features, N(0, 0.01) weights, zero bias -- same module tree
    (``linear.weight`` / ``linear.bias``), so the reference's probe checkpoints (checkpoint.pth.tar: "state_dict") load"""

    def __init__(self, dim, num_labels=1000):
        super().__init__()
        self.linear = torch.nn.Linear(dim, num_labels)
        self.linear.weight.data.normal_(mean=0.0, std=0.01)
        self.linear.bias.data.zero_()

    def forward(self, x):
        x = x.view(x.size(0), -1)
        if x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and self.linear.out_features % 4 == 0:  # (the fp32 GEMM's 16-byte rows)
            return _ProbeLinearFn.apply(x, self.linear.weight, self.linear.bias)
        return self.linear(x)  # (host tensors: the CPU tests of the loop logic)


def accuracy(output, target, topk=(1,)):
    """utils.py:559-566"""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0) * 100.0 / target.size(0) for k in topk]


def _features(model, inp, n, avgpool, depths):
    with torch.no_grad():
        return model.forward_return_n_last_blocks(inp, n, avgpool, depths).float()


def train_linear_epoch(model, linear_classifier, optimizer, loader, epoch, n, avgpool, depths):
    """eval_linear.train (eval_linear.py:244-277): frozen backbone features (forward_return_n_last_blocks, HIP path) -> linear
    classifier -> cross-entropy -> SGD step.  Returns the epoch means the reference logs ({"loss", "lr"}), rank-averaged."""
    linear_classifier.train()
    dev = next(linear_classifier.parameters()).device
    loss_sum, lr_sum, count = torch.zeros((), device=dev), 0.0, 0
    for inp, target in loader:
        inp, target = inp.to(dev, non_blocking=True), target.to(dev, non_blocking=True)
        output = linear_classifier(_features(model, inp, n, avgpool, depths))
        loss = torch.nn.functional.cross_entropy(output, target)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        loss_sum += loss.detach()
        lr_sum += optimizer.param_groups[0]["lr"]
        count += 1
    stats = torch.stack([loss_sum, torch.tensor(float(count), device=dev)])
    if _dist_on():
        dist.all_reduce(stats)
    return {"loss": (stats[0] / stats[1].clamp(min=1)).item(), "lr": lr_sum / max(count, 1)}


@torch.no_grad()
def validate_network(val_loader, model, linear_classifier, n, avgpool, depths):
    """eval_linear.validate_network (eval_linear.py:280-304) -> {"loss", "acc1", "acc5"}: loss averaged per batch, accuracies per
    sample, as the reference's MetricLogger does"""
    linear_classifier.eval()
    dev = next(linear_classifier.parameters()).device
    acc = torch.zeros(5, device=dev)  # sum of batch losses, batches, sum acc1 * bs, sum acc5 * bs, samples
    for inp, target in val_loader:
        inp, target = inp.to(dev, non_blocking=True), target.to(dev, non_blocking=True)
        output = linear_classifier(_features(model, inp, n, avgpool, depths))
        loss = torch.nn.functional.cross_entropy(output, target)
        a1, a5 = accuracy(output, target, topk=(1, min(5, output.shape[1])))
        bs = inp.shape[0]
        acc += torch.stack([loss, torch.ones((), device=dev), a1 * bs, a5 * bs, torch.tensor(float(bs), device=dev)])
    if _dist_on():
        dist.all_reduce(acc)
    return {"loss": (acc[0] / acc[1].clamp(min=1)).item(), "acc1": (acc[2] / acc[4].clamp(min=1)).item(), "acc5": (acc[3] / acc[4].clamp(min=1)).item()}
