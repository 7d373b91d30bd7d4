"""The checkpoint-FILE branch of the three plugins (what a user with the real weights runs), without
INKLAYER_RANDOM_WEIGHTS: seeded state dicts are written to temp files in the reference's own formats

  * GroundingDINO: {"model": state_dict} with a "module." prefix on every key, BERT (`bert.*`) and `feat_map.*`
    included, loaded with strict=False semantics            (GD/util/inference.py:29-36, GD/util/misc.py:711-717)
  * SAM: flat state_dict                                   (SA/build_sam.py:103-106)
  * Depth-Anything-V2 vitb: flat state_dict                (InkLayer/refinement/depth_sort.py:35-40)

and the plugins - pointed at those files through their own module attributes / arguments - must return results
bit-identical to engines built from the same dicts in memory.  GPU box only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bert_state_dict():
    """bert-base-uncased's ARCHITECTURE with seeded random weights (the real ones are not available offline)."""
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    m = BertModel(BertConfig(), add_pooling_layer=False).eval()
    return {"bert." + k: v.detach().clone() for k, v in m.state_dict().items()}


@torch.no_grad()
def test_plugins_load_reference_format_checkpoint_files(dev, tmp_path, monkeypatch):
    pytest.importorskip("transformers")
    from PIL import Image
    monkeypatch.delenv("INKLAYER_RANDOM_WEIGHTS", raising=False)
    import InkLayer.detector.gdino as DET
    import InkLayer.segmentor.sam as SEG
    import InkLayer.refinement.depth_sort as DS
    from InkLayer.utils import processing as P
    from inklayer_amd import depth, gdino, ops, sam, synthetic, text_branch, weights_init

    W, H = 640, 480
    rgb = synthetic.synthetic_sketch(3, H, W)
    png = tmp_path / "sketch.png"
    Image.fromarray(rgb).save(png)

    # ------------------------------------------------------------------ detector
    gcfg = gdino.GDinoConfig()
    gsd = {k: v.cpu() for k, v in weights_init.random_gdino_state_dict(gcfg, dev, 21).items()}
    for leaf in ("weight", "bias"):       # spread the scores (random weights saturate them) so that the 0.2 threshold cuts
        gsd[f"transformer.decoder.norm.{leaf}"] = gsd[f"transformer.decoder.norm.{leaf}"] * 0.05
    gsd.update(_bert_state_dict())
    torch.manual_seed(1)
    gsd["feat_map.weight"], gsd["feat_map.bias"] = torch.randn(256, 768) * 0.03, torch.randn(256) * 0.1
    gfile = tmp_path / "inklayer_gdino.pth"
    torch.save({"model": {"module." + k: v for k, v in gsd.items()}}, gfile)
    text = text_branch.encode_caption_from_checkpoint(gsd, gdino.DEFAULT_TOKEN_IDS)
    ref_eng = gdino.GDinoEngine(gsd, gcfg, dev, encoded_text=text)
    raw = torch.from_numpy(np.ascontiguousarray(rgb)).to(dev)
    oh, ow = gdino.resize_shape(W, H)
    ref_boxes, ref_scores = ref_eng.detect([ops.resize_bilinear_u8(raw, oh, ow)])[0]
    del ref_eng
    monkeypatch.setattr(DET, "weights_path", str(gfile))
    monkeypatch.setattr(DET, "model", None)
    out = DET.run_ft_dino_on_sketch(str(png))
    n = len(out["bboxes"])
    print("checkpoint-file detector keeps", n, "boxes; scores", out["scores"][:5])
    assert n > 0
    assert out["scores"] == ref_scores.tolist() and out["labels"] == ["object"] * n
    assert np.array_equal(np.asarray(out["bboxes"]), P.cxcywh_to_xyxy(ref_boxes.tolist()))
    monkeypatch.setattr(DET, "weights_path", str(tmp_path / "missing.pth"))
    monkeypatch.setattr(DET, "model", None)
    with pytest.raises(FileNotFoundError):
        DET.run_ft_dino_on_sketch(str(png))
    monkeypatch.setattr(DET, "model", None)
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ segmentor
    scfg = sam.SamConfig()
    ssd = {k: v.cpu() for k, v in weights_init.random_sam_state_dict(scfg, dev, 22).items()}
    sfile = tmp_path / "sam_vit_h_4b8939.pth"
    torch.save(ssd, sfile)
    boxes = torch.tensor([[30.0, 40.0, 400.0, 420.0], [200.0, 100.0, 630.0, 470.0], [5.0, 300.0, 300.0, 475.0]])
    ref_eng = sam.SamEngine(ssd, scfg, dev)
    want = sam.run_SAM(Image.fromarray(rgb), boxes, engine=ref_eng)
    del ref_eng
    monkeypatch.setattr(SEG, "_engine", None)
    got = SEG.run_SAM(Image.fromarray(rgb), boxes, sam_checkpoint=str(sfile))
    assert len(got) == 3 and all(np.array_equal(g, w) for g, w in zip(got, want))
    assert 0.02 < np.mean([g.mean() for g in got]) < 0.98
    again = SEG.run_SAM(Image.fromarray(rgb), boxes, sam_checkpoint=str(sfile))          # cached engine, not a reload
    assert all(np.array_equal(g, w) for g, w in zip(again, want)) and str(sfile) in sam._ENGINES
    with pytest.raises(FileNotFoundError):
        SEG.run_SAM(Image.fromarray(rgb), boxes, sam_checkpoint=str(tmp_path / "missing.pth"))
    sam._ENGINES.pop(str(sfile), None)
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ depth model
    dcfg = depth.DepthConfig()
    dsd = {k: v.cpu() for k, v in weights_init.random_depth_state_dict(dcfg, dev, 23).items()}
    dfile = tmp_path / "depth_anything_v2_vitb.pth"
    torch.save(dsd, dfile)
    ref_eng = depth.DepthEngine(dsd, dcfg, dev)
    want_d = ref_eng.infer_image(np.ascontiguousarray(rgb[..., ::-1])).cpu().numpy()
    del ref_eng
    monkeypatch.setattr(DS, "_engine", None)
    monkeypatch.setattr(DS, "get_model_path", lambda name: str(tmp_path / name))
    got_d = DS.get_depth_map(str(png))
    assert got_d.shape == (H, W) and np.array_equal(got_d, want_d) and float(got_d.std()) > 0
    monkeypatch.setattr(DS, "get_model_path", lambda name: str(tmp_path / "nowhere" / name))
    monkeypatch.setattr(DS, "_engine", None)
    with pytest.raises(FileNotFoundError):
        DS.get_depth_map(str(png))
    monkeypatch.setattr(DS, "_engine", None)
    depth._ENGINES.pop(str(dfile), None)
    torch.cuda.empty_cache()
