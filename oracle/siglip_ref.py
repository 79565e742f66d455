"""ORACLE (test infrastructure): plain-torch CPU restatement of the SigLIP vision tower as the
reference uses it -- functions over the HF-style state dict (`vision_model.*` keys).

Follows models/multimodal_encoder/siglip_encoder.py: SigLipVisionEmbeddings.forward :173-178,
SigLipAttention.forward :201-243 (fp32 softmax), SigLipMLP :256-259 (gelu_pytorch_tanh),
SigLipEncoderLayer.forward :296-309, and SigLipVisionTower.load_model/forward :566-590: the LAST
encoder layer is deleted and `hidden_states[-1]` (output of the last kept layer, before
post_layernorm) is returned.  fp32 throughout (the tower is frozen and outside autocast)."""
import torch
import torch.nn.functional as F


def siglip_tower_ref(sd, images, *, num_layers_total, num_heads, patch, eps=1e-6, prefix="vision_model."):
    p = prefix
    x = F.conv2d(images, sd[p + "embeddings.patch_embedding.weight"], sd[p + "embeddings.patch_embedding.bias"], stride=patch)
    h = x.flatten(2).transpose(1, 2) + sd[p + "embeddings.position_embedding.weight"][None]
    B, T, D = h.shape
    hd = D // num_heads
    for i in range(num_layers_total - 1):                      # tower drops the last layer
        q = p + f"encoder.layers.{i}."
        r = h
        y = F.layer_norm(h, (D,), sd[q + "layer_norm1.weight"], sd[q + "layer_norm1.bias"], eps)
        qs = F.linear(y, sd[q + "self_attn.q_proj.weight"], sd[q + "self_attn.q_proj.bias"]).view(B, T, num_heads, hd).transpose(1, 2)
        ks = F.linear(y, sd[q + "self_attn.k_proj.weight"], sd[q + "self_attn.k_proj.bias"]).view(B, T, num_heads, hd).transpose(1, 2)
        vs = F.linear(y, sd[q + "self_attn.v_proj.weight"], sd[q + "self_attn.v_proj.bias"]).view(B, T, num_heads, hd).transpose(1, 2)
        w = torch.matmul(qs, ks.transpose(2, 3)) * hd ** -0.5
        w = F.softmax(w, dim=-1, dtype=torch.float32)
        a = torch.matmul(w, vs).transpose(1, 2).contiguous().reshape(B, T, D)
        h = r + F.linear(a, sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"])
        r = h
        y = F.layer_norm(h, (D,), sd[q + "layer_norm2.weight"], sd[q + "layer_norm2.bias"], eps)
        y = F.linear(y, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"])
        y = F.gelu(y, approximate="tanh")
        h = r + F.linear(y, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
    return h


def siglip_param_shapes(hidden, inter, layers, channels, patch, image):
    out = [("vision_model.embeddings.patch_embedding.weight", (hidden, channels, patch, patch)),
           ("vision_model.embeddings.patch_embedding.bias", (hidden,)),
           ("vision_model.embeddings.position_embedding.weight", ((image // patch) ** 2, hidden))]
    for i in range(layers):
        q = f"vision_model.encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            out += [(q + n + ".weight", (hidden,)), (q + n + ".bias", (hidden,))]
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out += [(q + f"self_attn.{n}.weight", (hidden, hidden)), (q + f"self_attn.{n}.bias", (hidden,))]
        out += [(q + "mlp.fc1.weight", (inter, hidden)), (q + "mlp.fc1.bias", (inter,)),
                (q + "mlp.fc2.weight", (hidden, inter)), (q + "mlp.fc2.bias", (hidden,))]
    out += [("vision_model.post_layernorm.weight", (hidden,)), ("vision_model.post_layernorm.bias", (hidden,))]
    return out
