"""Row T (text branch): inklayer_amd.text_branch.bert_encode against the HuggingFace BertModel it restates.

The reference runs BertModelWarper.forward (GD/models/GroundingDINO/bertwarper.py:31-166) = HF BertModel's embeddings +
encoder with (a) the sub-sentence block attention mask and (b) explicit position ids from
generate_masks_with_special_tokens_and_transfer_map (:224-273).  bert-base-uncased itself is not available offline, so
the pin is architectural: a seeded random BertModel of the same shape family (768 hidden, 12 heads) is run by
`transformers` with exactly those two inputs and its state_dict is pushed through our restatement.  What stays DATA is
only the vocabulary id of "object" (4874, unverifiable offline)."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf(n_layers):
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    cfg = BertConfig(num_hidden_layers=n_layers, vocab_size=30522, hidden_size=768, num_attention_heads=12,
                     intermediate_size=3072, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():                       # make LayerNorm affine / biases non-trivial
        for n, p in m.named_parameters():
            if n.endswith("LayerNorm.weight"):
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
    return m


@pytest.mark.parametrize("token_ids", [[101, 4874, 1012, 102], [101, 4874, 3899, 1012, 4937, 1012, 102]])
def test_bert_encode_matches_hf_bert_with_block_mask_and_position_ids(token_ids):
    from inklayer_amd import gdino, text_branch
    m = _hf(2)
    mask, pos = gdino.text_masks_and_position_ids(token_ids)
    ids = torch.tensor([token_ids])
    add = torch.zeros(mask.shape, dtype=torch.float32).masked_fill(~mask, torch.finfo(torch.float32).min)
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=add[None, None], position_ids=pos[None],
                token_type_ids=torch.zeros_like(ids)).last_hidden_state[0]
    sd = {"bert." + k: v for k, v in m.state_dict().items()}
    got = text_branch.bert_encode(sd, token_ids)
    err = (got - ref).abs().max().item()
    print("bert_encode vs HF BertModel: max abs diff", err)
    assert err < 5e-6
    # the block mask matters (a plain all-ones mask gives a different answer), so the test is not vacuous
    with torch.no_grad():
        plain = m(input_ids=ids, position_ids=pos[None]).last_hidden_state[0]
    assert (plain - ref).abs().max().item() > 1e-3


def test_encode_caption_accepts_module_prefixed_checkpoint():
    """GroundingDINO checkpoints ship 'module.'-prefixed keys; the reference always runs clean_state_dict
    (GD/util/misc.py:711-717, GD/util/inference.py:33-34) before use."""
    from inklayer_amd import gdino, text_branch
    m = _hf(1)
    sd = {"module.bert." + k: v for k, v in m.state_dict().items()}
    torch.manual_seed(1)
    sd["module.feat_map.weight"] = torch.randn(256, 768) * 0.03
    sd["module.feat_map.bias"] = torch.randn(256) * 0.1
    clean = gdino.clean_state_dict(sd)
    assert all(not k.startswith("module.") for k in clean)
    out = text_branch.encode_caption_from_checkpoint(clean, gdino.DEFAULT_TOKEN_IDS)
    assert out.shape == (4, 256) and torch.isfinite(out).all()
    ref = torch.nn.functional.linear(text_branch.bert_encode(clean, gdino.DEFAULT_TOKEN_IDS),
                                     sd["module.feat_map.weight"], sd["module.feat_map.bias"])
    assert torch.equal(out, ref)
    with pytest.raises(KeyError):               # the raw (un-cleaned) dict is what used to blow up at load
        text_branch.encode_caption_from_checkpoint(sd, gdino.DEFAULT_TOKEN_IDS)
