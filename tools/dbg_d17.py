import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import gdino_ref, sam_ref
from inklayer_amd import gdino
dev = torch.device("cuda:0")
oc = gdino_ref.GDinoConfig(enc_layers=2, dec_layers=2, num_queries=300)
sd = sam_ref.seeded_state_dict(gdino_ref.gdino_param_shapes(oc), 77)
for k in sd:
    if k.endswith("gamma_v") or k.endswith("gamma_l"):
        sd[k] = 0.3 * torch.ones_like(sd[k]) + 0.05 * sd[k]
rs = np.random.RandomState(3)
text = torch.from_numpy((0.5 * rs.standard_normal((4, 256))).astype(np.float32))
rs = np.random.RandomState(12)
img = rs.randint(0, 256, size=(224, 288, 3)).astype(np.uint8)
mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
x = ((torch.from_numpy(img).permute(2, 0, 1).float() / 255.0) - mean.view(3, 1, 1)) / std.view(3, 1, 1)
sm, pid = gdino_ref.text_masks_and_position_ids([101, 4874, 1012, 102])
for scale in (1.0, 0.05):
    sd2 = dict(sd)
    for leaf in ("weight", "bias"):
        sd2[f"transformer.decoder.norm.{leaf}"] = sd[f"transformer.decoder.norm.{leaf}"] * scale
    with torch.no_grad():
        st = {}
        rl, rb = gdino_ref.detector_forward(sd2, oc, x[None], text, sm, pid, stages=st)
        eng = gdino.GDinoEngine(sd2, gdino.GDinoConfig(enc_layers=2, dec_layers=2, num_queries=300), dev, encoded_text=text)
        for force in (True, False):
            est = {"force_topk": st["topk"]} if force else {}
            lg, bx = eng.forward([torch.from_numpy(img).to(dev)], stages=est)
            d = (lg[0].cpu() - rl[0])
            print(f"scale {scale} force_topk {force}: ref logit absmax {rl.abs().max():.3f}; diff mean {d.mean():.4f} std {d.std():.4f} absmax {d.abs().max():.4f}; per-token mean diff {d.mean(0).tolist()}")
            if not force:
                mine = est["topk_logits"].max(-1)[0][0].cpu()
                got = torch.sort(mine, descending=True, stable=True)[1][:300]
                print("   topk order equal positions:", (got == st["topk"][0]).float().mean().item())
