"""MaskGIT training-time masking on the device -- drop-in for the reference's `data/masking.py`
(`mask_or_random_replace_tokens`, :13-94; called once per step from training/train.py:533-547 and
train_w_clip_vit.py / train_dpo.py), same signature and return tuple.

The random draws stay torch's (`torch.rand` on the tokens' device, same order as the reference: timesteps [B], then
scores [B, n]), so a seeded run masks the same positions; the argsort + compare + two `torch.where`s are one kernel
(`ug_maskgit_train_mask`).  The two optional branches -- `eval_mask_ratios` at evaluation time and
`mask_contiguous_region_prob` -- draw from Python's `random` in the reference's order."""
import math
import random

import torch

from unigen_hip import ops
from unigen_hip.lib import UniGenHipError


def mask_or_random_replace_tokens(image_tokens, mask_id, config, mask_schedule, is_train=True):
    tr = config.training
    batch_size, seq_len = image_tokens.shape
    dev = image_tokens.device
    if not is_train and tr.get("eval_mask_ratios", None):
        # evaluation: one of the configured ratios per image, from Python's `random` like the reference (:20-22)
        mask_prob = torch.tensor(random.choices(tr.eval_mask_ratios, k=batch_size), device=dev)
    else:
        timesteps = torch.rand(batch_size, device=dev)
        mask_prob = mask_schedule(timesteps)
        mask_prob = mask_prob.clip(tr.min_masking_rate)
    num_token_masked = (seq_len * mask_prob).round().clamp(min=1)
    region_prob = tr.get("mask_contiguous_region_prob", None)
    if region_prob is not None and random.random() < region_prob:
        # one rectangle of ~num_token_masked positions per image (:41-66): the bounds come from Python's `random` in the
        # reference's order (height, then the two start indices, image by image), so a seeded run masks the same rectangle;
        # the rectangles are assembled on the host (B small integers) and applied on the device
        resolution = int(seq_len ** 0.5)
        rect = torch.zeros((batch_size, resolution, resolution), dtype=torch.bool)
        for b, k in enumerate(num_token_masked.tolist()):
            k = int(k)
            h = min(random.randint(math.ceil(k / resolution), min(resolution, k)), resolution)
            w = min(math.ceil(k / h), resolution)
            r0 = random.randint(0, resolution - h)
            c0 = random.randint(0, resolution - w)
            rect[b, r0:r0 + h, c0:c0 + w] = True
        mask = rect.reshape(batch_size, -1).to(dev)
        if mask.shape[1] != seq_len:                       # (the reference's reshape fails the same way for non-square n)
            raise UniGenHipError(f"mask_contiguous_region needs a square token grid (n = {seq_len})")
        input_ids = torch.where(mask, mask_id, image_tokens)
        masked_labels = torch.where(mask, image_tokens, -100)
    else:
        scores = torch.rand(batch_size, seq_len, device=dev)
        input_ids, masked_labels = ops.maskgit_train_mask(image_tokens, scores, num_token_masked, mask_id, -100)
    # the reference's `if config.training.get("noise_type", "mask"):` is true for every non-empty string, so the input is
    # always the mask-token form; only the label / loss-weight convention depends on the options
    if tr.get("predict_all_tokens", False) or tr.get("noise_type", "mask") == "random_replace":
        mask = (masked_labels != -100).long()
        labels = image_tokens
        loss_weight = 1 - (1 - mask) * ((1 - mask_prob) * (1 - 0.3))[:, None]
    else:
        labels, loss_weight = masked_labels, None
    return input_ids, labels, loss_weight, mask_prob
