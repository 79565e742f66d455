"""MaskGIT training-time masking on the device -- drop-in for the reference's `data/masking.py`
(`mask_or_random_replace_tokens`, :13-94; called once per step from training/train.py:533-547 and
train_w_clip_vit.py / train_dpo.py), same signature and return tuple.

The random draws stay torch's (`torch.rand` on the tokens' device, same order as the reference: timesteps [B], then
scores [B, n]), so a seeded run masks the same positions; the argsort + compare + two `torch.where`s are one kernel
(`ug_maskgit_train_mask`)."""
import torch

from unigen_hip import ops
from unigen_hip.lib import UniGenHipError


def mask_or_random_replace_tokens(image_tokens, mask_id, config, mask_schedule, is_train=True):
    tr = config.training
    if not is_train and tr.get("eval_mask_ratios", None):
        raise UniGenHipError("mask_or_random_replace_tokens: eval_mask_ratios is not used by any shipped config and is not implemented")
    if tr.get("mask_contiguous_region_prob", None) is not None:
        raise UniGenHipError("mask_or_random_replace_tokens: mask_contiguous_region_prob is not used by any shipped config "
                             "and is not implemented")
    batch_size, seq_len = image_tokens.shape
    dev = image_tokens.device
    timesteps = torch.rand(batch_size, device=dev)
    mask_prob = mask_schedule(timesteps)
    mask_prob = mask_prob.clip(tr.min_masking_rate)
    num_token_masked = (seq_len * mask_prob).round().clamp(min=1)
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
