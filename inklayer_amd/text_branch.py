"""Load-time constant folding of GroundingDINO's text branch (groundingdino.py:248-297).

InkLayer always prompts with the caption "object" (InkLayer/detector/gdino.py:18), so
`encoded_text = feat_map(BERT(tokens))` does not depend on the image: it is computed ONCE when a
checkpoint is loaded (plain torch on the host: ~0.7 GFLOP, never on the hot path) and handed to the
detector engine as a [n_tokens, 256] tensor.

Parity: this restatement of BertModelWarper.forward (GD/.../bertwarper.py:31-166: HF BertModel with the
sub-sentence block attention mask and explicit position ids) is pinned against `transformers.BertModel` itself on
seeded random weights of the same architecture (tests/test_text_branch_cpu.py, max abs diff ~1e-6).  The
bert-base-uncased WEIGHTS and VOCABULARY are not available offline: they arrive with the GroundingDINO checkpoint
(`bert.*` keys), and the token ids of the caption are DATA here ([CLS]=101, "object"=4874 (unverifiable offline),
"."=1012, [SEP]=102).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from .gdino import text_masks_and_position_ids


@torch.no_grad()
def bert_encode(sd: Dict[str, torch.Tensor], token_ids: Sequence[int], prefix: str = "bert.") -> torch.Tensor:
    """last_hidden_state [n, 768] of BERT-base for one sentence with the sub-sentence block mask."""
    g = lambda n: sd[prefix + n].detach().float().cpu()
    ids = torch.tensor(list(token_ids), dtype=torch.long)
    mask, pos_ids = text_masks_and_position_ids(list(token_ids))
    x = g("embeddings.word_embeddings.weight")[ids] + g("embeddings.token_type_embeddings.weight")[0] \
        + g("embeddings.position_embeddings.weight")[pos_ids]
    x = F.layer_norm(x, (x.shape[-1],), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), 1e-12)
    add_mask = (1.0 - mask.float()) * torch.finfo(torch.float32).min
    n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(prefix + "encoder.layer."))
    H = 12
    for i in range(n_layers):
        p = f"encoder.layer.{i}."
        lin = lambda t, n: F.linear(t, g(p + n + ".weight"), g(p + n + ".bias"))
        n, D = x.shape
        sp = lambda t: t.view(n, H, D // H).transpose(0, 1)
        q, k, v = sp(lin(x, "attention.self.query")), sp(lin(x, "attention.self.key")), sp(lin(x, "attention.self.value"))
        a = (q @ k.transpose(-1, -2)) / math.sqrt(D // H) + add_mask[None]
        ctx = (a.softmax(-1) @ v).transpose(0, 1).reshape(n, D)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (D,), g(p + "attention.output.LayerNorm.weight"),
                         g(p + "attention.output.LayerNorm.bias"), 1e-12)
        hmid = F.gelu(lin(x, "intermediate.dense"))
        x = F.layer_norm(lin(hmid, "output.dense") + x, (D,), g(p + "output.LayerNorm.weight"),
                         g(p + "output.LayerNorm.bias"), 1e-12)
    return x


@torch.no_grad()
def encode_caption_from_checkpoint(sd: Dict[str, torch.Tensor], token_ids: Sequence[int]) -> torch.Tensor:
    """feat_map(BERT(tokens)) -> [n, 256] f32 (groundingdino.py:277-279)."""
    hidden = bert_encode(sd, token_ids)
    return F.linear(hidden, sd["feat_map.weight"].detach().float().cpu(), sd["feat_map.bias"].detach().float().cpu())
