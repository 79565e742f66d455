"""Parity at the BENCHMARKED shapes: the fixtures stop at L = 387 / V = 377, the headline bench runs
L = 771 (13 key tiles), the SFT config L = 1603 with the mmu_vit mask, and the head + cross-entropy over V = 159 867.  Each
test runs ONE decoder layer of the 1.5B model's width on the HIP path and on the pinned CPU oracle (bf16 autocast = the
reference's training mode) with the same seeded weights and inputs, and prints the measured errors next to the gates:
losses <= 1e-3 relative (north_star), logits / gradients as relative Frobenius error."""
import pytest
import torch

from helpers import additive, fp32_yardstick, llm_config_dir, oracle_lm

pytestmark = pytest.mark.gpu
WIDE = dict(hidden_size=1536, intermediate_size=8960, num_hidden_layers=1, num_attention_heads=12, num_key_value_heads=2,
            rope_theta=1e6, rms_norm_eps=1e-6)
SEED = 71


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _pair(dev, V, text_vocab, codebook, nvq):
    from models import UniGen
    from oracle import weights
    cfg = dict(WIDE, vocab_size=V)
    m = UniGen(w_und_encoder=False, vocab_size=V, llm_vocab_size=text_vocab, llm_model_path=llm_config_dir(cfg), codebook_size=codebook,
               num_vq_tokens=nvq, load_from_pretrained=True, device=dev, init_seed=-1)
    names = [(n, tuple(p.shape)) for n, p in m.llm.named_parameters()]
    m.llm.load_state_dict(weights.synth_llm_state(names, seed=SEED), strict=False)
    lm, _ = oracle_lm(cfg, SEED)
    return m.train(), lm


def _report(tag, model, lm, logits_pair, loss_pair, grad_gate):
    (lg, lo), (l_hip, l_ref) = logits_pair, loss_pair
    lerr = abs(l_hip - l_ref) / abs(l_ref)
    rl = _rel(lg, lo)
    ref_g = dict(lm.named_parameters())
    worst, which = 0.0, None
    for n, p in model.llm.named_parameters():
        e = _rel(p.grad, ref_g[n].grad)
        if e > worst:
            worst, which = e, n
    print(f"[{tag}] loss {l_hip:.6f} vs oracle {l_ref:.6f} (rel {lerr:.2e}, gate 1e-3); logits rel {rl:.2e}; "
          f"worst gradient rel {worst:.2e} ({which}), gate {grad_gate:.0e}")
    assert lerr < 1e-3
    assert worst < grad_gate, (which, worst)
    return rl


def test_decoder_layer_L771_left_pad_matches_oracle(dev):
    """The headline shape: 513 text + 258 image-segment tokens; row 1 carries 25 % left padding (192 pads), so the pad / text /
    image-segment regions of the t2i mask all cross key-tile boundaries."""
    from oracle import host_ref, qwen2_ref
    TV, CB, n, L = 312, 64, 256, 771
    V = TV + CB + 1
    PAD, SOI, EOI, MASK = 300, 303, 304, V - 1
    model, lm = _pair(dev, V, TV, CB, n)
    g = torch.Generator().manual_seed(5)
    B = 2
    seq = torch.randint(0, 290, (B, L), generator=g)
    seq[1, :192] = PAD
    seq[:, -(n + 2)] = SOI
    seq[:, -1] = EOI
    img = torch.randint(TV, TV + CB, (B, n), generator=g)
    msk = torch.rand(B, n, generator=g) < torch.tensor([[0.9], [0.3]])
    msk[:, 0] = True
    seq[:, -(n + 1):-1] = torch.where(msk, MASK, img)
    labels = torch.full((B, L), -100)
    labels[:, -(n + 1):-1] = torch.where(msk, img, -100)
    mask = additive(host_ref.mask_predict_next_ref(seq, PAD, SOI, EOI, rm_pad_in_image=True))
    lo, r1, _, _ = qwen2_ref.unigen_forward_ref(lm, seq, mask, labels, batch_size_t2i=B, num_vq_tokens=n, autocast=True)
    r1.backward()
    logits, l1, _, _ = model(input_ids=seq.to(dev), attention_mask=mask.to(dev), labels=labels.to(dev), batch_size_t2i=B,
                             max_seq_length=L - n - 3, num_vq_tokens=n)
    model.llm.engine.check_errors()
    l1.backward()
    got = logits[:, -(n + 1):-1, :].float().cpu()
    rl = _report("L=771, 25% left pad", model, lm, (got, lo[:, -(n + 1):-1]), (l1.item(), r1.item()), 3e-2)
    assert rl < 1e-2
    top2 = lo[:, -(n + 1):-1].topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.05
    assert torch.equal(got.argmax(-1)[clear], lo[:, -(n + 1):-1].argmax(-1)[clear]) and clear.float().mean() > 0.5
    with torch.no_grad():
        lo32 = qwen2_ref.unigen_forward_ref(lm, seq, mask, None, autocast=False)[:, -(n + 1):-1]
    fp32_yardstick("L=771, 25% left pad", got, lo[:, -(n + 1):-1], lo32)


def test_decoder_layer_L1603_mmu_vit_mask_matches_oracle(dev):
    """The SFT understanding shape: 19 prefix ids | 729 projected image embeddings | 855 text ids = 1603 positions under
    create_attention_mask_for_mmu_vit (image columns visible to every row), embeddings in, shifted CE on the text, gradient
    w.r.t. the input embeddings (what reaches mm_projector) compared too."""
    from oracle import host_ref, qwen2_ref
    TV, CB, n = 312, 64, 256
    V = TV + CB + 1
    L, P, NI = 1603, 19, 729
    model, lm = _pair(dev, V, TV, CB, n)
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 290, (1, L), generator=g)
    with torch.no_grad():
        emb = lm.model.embed_tokens(ids).clone()
    emb[:, P:P + NI] = 0.02 * torch.randn(1, NI, WIDE["hidden_size"], generator=g)
    labels = ids.clone()
    labels[:, :P + NI + 1] = -100
    mask = additive(host_ref.mask_mmu_vit_ref(1, L, prefix_length=P, num_tokens=NI))
    e_ref = emb.clone().requires_grad_(True)
    lo, _, _, r3 = qwen2_ref.unigen_forward_ref(lm, None, mask, labels, input_embeddings=e_ref, batch_size_mmu=1, num_vq_tokens=n,
                                                autocast=True)
    r3.backward()
    e_hip = emb.clone().to(dev).requires_grad_(True)
    logits, _, _, l3 = model(input_ids=None, input_embeddings=e_hip, attention_mask=mask.to(dev), labels=labels.to(dev), batch_size_mmu=1,
                             max_seq_length=L - n - 3, num_vq_tokens=n)
    model.llm.engine.check_errors()
    l3.backward()
    rows = slice(P + NI, L - 1)
    # the embedding table receives no lookup gradient here (embeddings were passed in): only the tied head's part
    rl = _report("L=1603, mmu_vit mask", model, lm, (logits[:, rows, :].float().cpu(), lo[:, rows]), (l3.item(), r3.item()), 3e-2)
    de = _rel(e_hip.grad, e_ref.grad)
    print(f"[L=1603, mmu_vit mask] d(loss)/d(input_embeddings) rel {de:.2e}")
    assert rl < 1e-2 and de < 3e-2
    with torch.no_grad():
        lo32 = qwen2_ref.unigen_forward_ref(lm, None, mask, None, input_embeddings=emb, autocast=False)[:, rows]
    fp32_yardstick("L=1603, mmu_vit mask", logits[:, rows, :].float().cpu(), lo[:, rows], lo32)


def test_head_and_cross_entropy_full_vocabulary_matches_oracle(dev):
    """Tied lm_head + CE over the real vocabulary (159 867 = 151 674 text ids + 8 192 codes + mask id; 624.5 column tiles of
    256, ragged last tile): 64 label rows, loss / logits over all columns / the dense embedding gradient."""
    from oracle import host_ref, qwen2_ref
    TV, CB, n = 151674, 8192, 64
    V = TV + CB + 1
    PAD, SOI, EOI, MASK = 151643, 151665, 151666, V - 1
    model, lm = _pair(dev, V, TV, CB, n)
    g = torch.Generator().manual_seed(7)
    L = 16 + n + 3
    seq = torch.randint(0, 151643, (1, L), generator=g)
    seq[:, :5] = PAD
    seq[:, -(n + 2)] = SOI
    seq[:, -1] = EOI
    img = torch.randint(TV, TV + CB, (1, n), generator=g)
    seq[:, -(n + 1):-1] = MASK
    labels = torch.full((1, L), -100)
    labels[:, -(n + 1):-1] = img
    mask = additive(host_ref.mask_predict_next_ref(seq, PAD, SOI, EOI, rm_pad_in_image=True))
    lo, r1, _, _ = qwen2_ref.unigen_forward_ref(lm, seq, mask, labels, batch_size_t2i=1, num_vq_tokens=n, autocast=True)
    r1.backward()
    logits, l1, _, _ = model(input_ids=seq.to(dev), attention_mask=mask.to(dev), labels=labels.to(dev), batch_size_t2i=1,
                             max_seq_length=16, num_vq_tokens=n)
    model.llm.engine.check_errors()
    l1.backward()
    got = logits[:, -(n + 1):-1, :].float().cpu()
    assert got.shape == (1, n, V)
    rl = _report("V=159867 head + CE, 64 rows", model, lm, (got, lo[:, -(n + 1):-1]), (l1.item(), r1.item()), 3e-2)
    assert rl < 1e-2
    with torch.no_grad():
        lo32 = qwen2_ref.unigen_forward_ref(lm, seq, mask, None, autocast=False)[:, -(n + 1):-1]
    fp32_yardstick("V=159867 head", got, lo[:, -(n + 1):-1], lo32)
