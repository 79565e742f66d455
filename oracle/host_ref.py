"""ORACLE (test infrastructure): restatements of the reference's caller-side sequence/mask logic
(SURVEY.md §8 rows a4-a6), vectorised -- no per-sample Python loops -- so they also document the
closed forms a device-side builder has to reproduce.

  * attention-mask builders: training/prompting_utils.py:975-1020 (predict_next), :1023-1036 (mmu),
    :1038-1074 (mmu_vit).  Returned as boolean "may attend" [N, L, L]; `to_additive` gives the
    reference's 0 / iinfo(int64).min encoding.
  * MaskGIT training masking: data/masking.py:13-94 (default branch: noise_type 'mask', no
    contiguous-region masking, predict_all_tokens False).
  * t2i token layout: training/prompting_utils.py:59-111 for already-tokenised text.
"""
import torch

NEG = torch.iinfo(torch.int64).min


def to_additive(allow, dtype=torch.int64):
    """[N,L,L] bool -> [N,1,L,L] additive mask exactly as the builders return it (return_inverse_mask=True)."""
    inv = (~allow).to(torch.int64)
    inv = inv.masked_fill(inv.to(torch.bool), NEG)
    return inv.unsqueeze(1).to(dtype)


def mask_predict_next_ref(seq, pad_id, soi_id, eoi_id, rm_pad_in_image=False):
    N, L = seq.shape
    is_pad = seq == pad_id
    start = (seq == soi_id)
    end = (seq == eoi_id)
    in_img = (start.cumsum(1) > end.cumsum(1)) | start | end
    r = torch.arange(L)
    causal = (r[None, :] <= r[:, None])[None].expand(N, L, L)               # [N, row, col]
    allow = torch.where(in_img[:, :, None], torch.ones((), dtype=torch.bool), causal)
    if rm_pad_in_image:
        has_pad = is_pad.any(1)
        last_pad = torch.where(has_pad, (is_pad * r[None, :]).max(1).values, torch.full((N,), -1))
        # text rows after the last pad never look at columns up to it
        cut = (r[None, :, None] > last_pad[:, None, None]) & (r[None, None, :] <= last_pad[:, None, None])
        allow = allow & ~(cut & ~in_img[:, :, None])
        # rows from <soi> on never look at pad columns
        soi_pos = start.float().argmax(1)
        from_soi = r[None, :] >= soi_pos[:, None]
        allow = allow & ~(in_img[:, :, None] & from_soi[:, :, None] & is_pad[:, None, :])
    return allow


def mask_mmu_ref(seq, eoi_id):
    N, L = seq.shape
    r = torch.arange(L)
    eoi_pos = torch.where(seq == eoi_id)[1][0]          # reference uses the FIRST match of the whole batch
    allow = (r[None, :] <= r[:, None]) | (r[None, :] <= eoi_pos)
    return allow[None].expand(N, L, L).clone()


def mask_mmu_vit_ref(N, L, prefix_length=-1, system_prompt_len=0, num_images=1, num_tokens=576):
    r = torch.arange(L)
    start = prefix_length if prefix_length > 0 else 1 + system_prompt_len + 1
    endi = start + num_tokens * num_images
    allow = (r[None, :] <= r[:, None]) | ((r[None, :] >= start) & (r[None, :] < endi))
    return allow[None].expand(N, L, L).clone()


def maskgit_train_mask_ref(image_tokens, mask_id, timesteps, rand_scores, schedule, min_masking_rate=0.0):
    """data/masking.py:13-94 with the two random draws injected: timesteps ~ U(0,1) [B] and
    rand_scores ~ U(0,1) [B, N] (the tensor whose argsort picks the masked positions)."""
    B, n = image_tokens.shape
    mask_prob = schedule(timesteps).clip(min_masking_rate)
    num_masked = (n * mask_prob).round().clamp(min=1)
    perm_rank = rand_scores.argsort(dim=-1)
    mask = perm_rank < num_masked.unsqueeze(-1)
    input_ids = torch.where(mask, mask_id, image_tokens)
    labels = torch.where(mask, image_tokens, -100)
    return input_ids, labels, mask_prob


def t2i_layout_ref(text_ids, image_ids, labels, max_seq_len, pad_id, soi_id, eoi_id, conv_start, conv_end,
                   ignore_id=-100):
    """t2i_prompt (prompting_utils.py:59-111) without the random prompt dropout: left-padded
    [pad.. | conv_start text conv_end | soi image eoi]; returns (ids, attn01, labels)."""
    n = image_ids.shape[1]
    seqs, masks, labs = [], [], []
    for i, t in enumerate(text_ids):
        body = list(conv_start) + list(t) + list(conv_end)
        room = max_seq_len - n - 2
        if room >= len(body):
            m = [0] * (room - len(body)) + [1] * (len(body) + n + 2)
            body = [pad_id] * (room - len(body)) + body
        else:
            m = [1] * max_seq_len
            body = body[:room]
        ids = torch.cat([torch.tensor(body, dtype=torch.long), torch.tensor([soi_id]), image_ids[i], torch.tensor([eoi_id])])
        lab = torch.cat([torch.full((len(body),), ignore_id, dtype=torch.long), torch.tensor([soi_id]), labels[i],
                         torch.tensor([eoi_id])])
        lab = torch.where(lab == pad_id, ignore_id, lab)
        seqs.append(ids); masks.append(torch.tensor(m)); labs.append(lab)
    return torch.stack(seqs), torch.stack(masks), torch.stack(labs)


def batch_logps_ref(logits, labels, num_vq_tokens, average_log_prob=False, label_pad_token_id=-100, t2i_gen_mode="mask"):
    """get_batch_logps (training/train_dpo.py:51-90): log-softmax over V at the image positions [-(n+1):-1], gather the
    label's log-prob, sum (or mean) over positions whose label is not the pad id; 'ar' mode shifts by one."""
    n = num_vq_tokens
    lg = logits[:, -(n + 1):-1]
    lb = labels[:, -(n + 1):-1].clone()
    keep = lb != label_pad_token_id
    lb[~keep] = 0
    if t2i_gen_mode == "ar":
        per = torch.gather(lg[:, :-1].log_softmax(-1), 2, lb[:, 1:].unsqueeze(2)).squeeze(2)
        keep = keep[:, 1:]
    else:
        per = torch.gather(lg.log_softmax(-1), 2, lb.unsqueeze(2)).squeeze(2)
    tot = (per * keep).sum(-1)
    return tot / keep.sum(-1) if average_log_prob else tot


def dpo_loss_ref(policy_chosen, policy_rejected, ref_chosen, ref_rejected, beta, coef=1.0):
    """training/train_dpo.py:640-647: coef * mean(-logsigmoid(beta * ((pi_c - pi_r) - (ref_c - ref_r))))."""
    import torch.nn.functional as F
    return coef * (-F.logsigmoid(beta * ((policy_chosen - policy_rejected) - (ref_chosen - ref_rejected)))).mean()
