"""HIP GroundingDINO path vs the CPU oracle (oracle/gdino_ref.py, pinned to the reference by
tests/golden/gdino_small.npz).  GPU box only."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden" / "gdino_small.npz"


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item(), ((a - b).norm() / b.norm()).item()


def test_ms_deform_attn_forward_reference_abi(dev):
    """The reference's own op signature (vision.cpp:53-56) against the golden produced by the
    reference's CPU form, and against the oracle at decoder-like shapes with out-of-range samples."""
    from inklayer_amd import ops
    from oracle import gdino_ref
    g = np.load(GOLD)
    shapes = [tuple(int(v) for v in s) for s in g["msda_shapes"]]
    starts = np.cumsum([0] + [a * b for a, b in shapes])[:-1]
    out = ops.ms_deform_attn_forward(torch.from_numpy(g["msda_value"]).to(dev), torch.tensor(shapes),
                                     torch.tensor(starts), torch.from_numpy(g["msda_loc"]).to(dev),
                                     torch.from_numpy(g["msda_w"]).to(dev), 64)
    assert (out.cpu() - torch.from_numpy(g["msda_out"])).abs().max().item() < 2e-5
    rs = np.random.RandomState(0)
    shapes = [(100, 100), (50, 50), (25, 25), (13, 13)]
    S = sum(a * b for a, b in shapes)
    v = torch.from_numpy(rs.standard_normal((2, S, 8, 32)).astype(np.float32))
    loc = torch.from_numpy(rs.uniform(-0.1, 1.1, size=(2, 300, 8, 4, 4, 2)).astype(np.float32))
    # exact corner / border cases: 0, 1, pixel centres
    loc[0, 0] = 0.0
    loc[0, 1] = 1.0
    loc[0, 2] = 0.5
    aw = torch.from_numpy(rs.uniform(0, 1, size=(2, 300, 8, 4, 4)).astype(np.float32))
    ref = gdino_ref.msda_core(v, shapes, loc, aw)
    starts = np.cumsum([0] + [a * b for a, b in shapes])[:-1]
    out = ops.ms_deform_attn_forward(v.to(dev), torch.tensor(shapes), torch.tensor(starts), loc.to(dev), aw.to(dev))
    assert (out.cpu() - ref).abs().max().item() < 3e-5
    with pytest.raises(Exception):   # B % min(B, im2col_step) != 0 is rejected like the reference (.cu:53)
        ops.ms_deform_attn_forward(torch.zeros(3, S, 8, 32, device=dev), torch.tensor(shapes), torch.tensor(starts),
                                   torch.zeros(3, 1, 8, 4, 4, 2, device=dev), torch.zeros(3, 1, 8, 4, 4, device=dev), 2)


def test_topk_rowmax(dev):
    from inklayer_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 13294, 4, generator=g)
    x[0, 100:140] = 0.25            # a block of exact ties -> must come out in index order
    x[1, 5] = float("inf")
    idx, val = ops.topk_rowmax(x.to(dev), 900, want_values=True)
    key = x.max(-1)[0]
    ref = torch.sort(key, dim=1, descending=True, stable=True)
    assert torch.equal(idx.cpu().long(), ref[1][:, :900])
    assert torch.equal(val.cpu(), ref[0][:, :900])
    # > 16384 tokens (800x1333 inputs: 22223): two-level path, incl. ties that straddle the chunk boundary
    y = torch.randn(2, 22223, 4, generator=g)
    y[0, 11000:11300] = 3.5
    idx = ops.topk_rowmax(y.to(dev), 900)
    ref = torch.sort(y.max(-1)[0], dim=1, descending=True, stable=True)
    assert torch.equal(idx.cpu().long(), ref[1][:, :900])


def test_fusion_fewkeys_groupnorm_ops(dev):
    from inklayer_amd import ops
    g = torch.Generator().manual_seed(2)
    B, S, T, E = 2, 1129, 4, 1024
    qv = torch.randn(B * S, 2 * E, generator=g).half()
    kl = torch.randn(B * T, 2 * E, generator=g).half()
    ov, ol = ops.biattn_fusion(qv.to(dev), kl.to(dev), B, S, T, 256 ** -0.5)
    q = qv[:, :E].double().view(B, S, 4, 256).transpose(1, 2)
    vv = qv[:, E:].double().view(B, S, 4, 256).transpose(1, 2)
    k = kl[:, :E].double().view(B, T, 4, 256).transpose(1, 2)
    vl = kl[:, E:].double().view(B, T, 4, 256).transpose(1, 2)
    aw = (q @ k.transpose(-1, -2)) * 256 ** -0.5
    rv = (aw.softmax(-1) @ vl).transpose(1, 2).reshape(B * S, E)
    rl = (aw.transpose(-1, -2).softmax(-1) @ vv).transpose(1, 2).reshape(B * T, E)
    assert (ov.double().cpu() - rv).abs().max().item() < 3e-3
    assert (ol.double().cpu() - rl).abs().max().item() < 3e-3
    # few-key attention with a block mask
    nq, nk, H, hd = 900, 4, 8, 32
    q = torch.randn(B * nq, H * hd, generator=g).half()
    k = torch.randn(B * nk, H * hd, generator=g).half()
    v = torch.randn(B * nk, H * hd, generator=g).half()
    o = ops.attn_fewkeys(q.to(dev), k.to(dev), v.to(dev), B=B, n_heads=H, head_dim=hd, scale=hd ** -0.5)
    qd, kd, vd = (t.double().view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    r = ((qd @ kd.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ vd
    assert (o.double().cpu() - r.transpose(1, 2).reshape(B * nq, H * hd)).abs().max().item() < 2e-3
    blocked = torch.tensor([[0, 1, 1, 1], [1, 0, 0, 1], [1, 0, 0, 1], [1, 1, 1, 0]], dtype=torch.uint8)
    q4 = torch.randn(B * 4, 256, generator=g).half()
    o = ops.attn_fewkeys(q4.to(dev), q4.to(dev), q4.to(dev), B=B, n_heads=4, head_dim=64, scale=0.125,
                         blocked=blocked.to(dev))
    qd = q4.double().view(B, 4, 4, 64).transpose(1, 2)
    a = (qd @ qd.transpose(-1, -2)) * 0.125
    a = a.masked_fill(blocked.bool()[None, None], float("-inf")).softmax(-1) @ qd
    assert (o.double().cpu() - a.transpose(1, 2).reshape(B * 4, 256)).abs().max().item() < 2e-3
    # GroupNorm(32, 256) on NHWC tokens, written at a batch stride
    x = torch.randn(B * 300, 256, generator=g) * 2 + 0.3
    gm, bt = torch.randn(256, generator=g), torch.randn(256, generator=g)
    out = torch.zeros(B * 500, 256, device=dev)
    ops.groupnorm_nhwc(x.to(dev), B, 300, 32, gm.to(dev), bt.to(dev), 1e-5, out[100:], 500 * 256)
    r = torch.nn.functional.group_norm(x.view(B, 300, 256).transpose(1, 2).double(), 32, gm.double(), bt.double(), 1e-5)
    got = out.view(B, 500, 256)[:, 100:400].cpu().double()
    assert (got - r.transpose(1, 2)).abs().max().item() < 1e-4


@pytest.fixture(scope="module")
def small_dino(dev):
    from oracle import gdino_ref, sam_ref
    from inklayer_amd import gdino
    oc = gdino_ref.GDinoConfig(enc_layers=2, dec_layers=2, num_queries=300)
    ec = gdino.GDinoConfig(enc_layers=2, dec_layers=2, num_queries=300)
    sd = sam_ref.seeded_state_dict(gdino_ref.gdino_param_shapes(oc), 77)
    for k in sd:
        if k.endswith("gamma_v") or k.endswith("gamma_l"):
            sd[k] = 0.3 * torch.ones_like(sd[k]) + 0.05 * sd[k]
    rs = np.random.RandomState(3)
    text = torch.from_numpy((0.5 * rs.standard_normal((4, 256))).astype(np.float32))
    eng = gdino.GDinoEngine(sd, ec, dev, encoded_text=text)
    return sd, oc, eng, text


@torch.no_grad()
def test_detector_stages_match_oracle(dev, small_dino):
    from oracle import gdino_ref
    from inklayer_amd import gdino
    sd, oc, eng, text = small_dino
    rs = np.random.RandomState(4)
    img = rs.randint(0, 256, size=(300, 412, 3)).astype(np.uint8)     # odd stage sizes: pads + odd merges
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    x = ((torch.from_numpy(img).permute(2, 0, 1).float() / 255.0) - mean.view(3, 1, 1)) / std.view(3, 1, 1)
    sm, pid = gdino_ref.text_masks_and_position_ids(list(gdino.DEFAULT_TOKEN_IDS))
    st = {}
    ref_logits, ref_boxes = gdino_ref.detector_forward(sd, oc, x[None], text, sm, pid, stages=st)
    est = {"force_topk": st["topk"]}
    logits, boxes = eng.forward([torch.from_numpy(img).to(dev)], stages=est)
    for i, f in zip((1, 2, 3), st["feats"]):
        ref = f[0].permute(1, 2, 0).reshape(-1, f.shape[1])
        mx, l2 = _rel(est["feats"][i][0], ref)
        print(f"swin stage {i}: max-rel {mx:.2e} l2-rel {l2:.2e}")
        assert mx < 2e-2 and l2 < 4e-3
    mx, l2 = _rel(est["src"], st["src"][0])
    print(f"input_proj src: max-rel {mx:.2e} l2-rel {l2:.2e}")
    assert mx < 2e-2 and l2 < 4e-3
    mx, l2 = _rel(est["memory"], st["memory"][0])
    print(f"encoder memory: max-rel {mx:.2e} l2-rel {l2:.2e}")
    assert mx < 3e-2 and l2 < 5e-3
    mx, l2 = _rel(est["memory_text"], st["memory_text"][0])
    print(f"memory_text: max-rel {mx:.2e} l2-rel {l2:.2e}")
    assert mx < 2e-2 and l2 < 5e-3
    # query selection: the unforced top-k agrees except where logits are within f16 noise of each other
    key = st["topk_logits"][0]
    mine = est["topk_logits"].max(-1)[0][0].cpu()
    print("topk logits max-rel", _rel(mine, key)[0])
    got = set(torch.sort(mine, descending=True, stable=True)[1][:300].tolist())
    want = set(st["topk"][0].tolist())
    print("topk set overlap", len(got & want) / 300)
    assert len(got & want) >= 0.97 * 300
    mx, l2 = _rel(est["ref0"], st["refs"][0][0])
    assert mx < 1e-2
    # decoder outputs per query: with RANDOM weights ~10 % of the queries are ill-conditioned (large random
    # sampling offsets on a random feature map; even two fp32 runs differ by 1e-3 there, see
    # tests/test_oracle_gdino.py), so the median / 75th percentile are held tight and the tail loose.
    d = (boxes[0].cpu() - ref_boxes[0]).abs().max(-1)[0]
    print("box err p50/p75/p90/max", d.median().item(), d.quantile(0.75).item(), d.quantile(0.9).item(), d.max().item())
    assert d.median().item() < 1e-3 and d.quantile(0.75).item() < 3e-3 and d.quantile(0.9).item() < 2e-2
    dl = (logits[0].cpu() - ref_logits[0]).abs().max(-1)[0] / ref_logits.abs().max()
    print("logit err p50/p75/p90/max", dl.median().item(), dl.quantile(0.75).item(), dl.quantile(0.9).item(), dl.max().item())
    assert dl.median().item() < 3e-3 and dl.quantile(0.75).item() < 8e-3 and dl.quantile(0.9).item() < 5e-2
    # MAX bound, justified by the oracle's OWN sensitivity to the stated arithmetic: the fp32 oracle is re-run with the
    # operands of every linear / conv rounded to f16 (exactly what DESIGN.md §4 says the HIP path does; weights and
    # activations, f32 accumulation) and, separately, with f16 weights only, with the query selection pinned.  With
    # RANDOM weights ~10 % of the queries are ill-conditioned (large random sampling offsets on a random feature map)
    # and move by up to several 1e-2 under that rounding in the fp32 oracle itself.  So
    #   (1) every quantile of the HIP error INCLUDING THE MAXIMUM is bounded by 2x the emulated-f16 oracle's, and
    #   (2) per query: err(q) <= 2e-3 + 20 * sens(q), sens = the query's own movement under the probes, for all but
    #       2 % of the queries (a different rounding realisation can hit a query the probes happened to miss).
    import torch.nn.functional as RealF

    class _F16Operands:
        def __getattr__(self, k):
            return getattr(RealF, k)

        def linear(self, a, w, b=None):
            return RealF.linear(a.half().float(), w.half().float(), b)

        def conv2d(self, a, w, b=None, **kw):
            return RealF.conv2d(a.half().float(), w.half().float(), b, **kw)

    def probe(sd_, proxy):
        pst = {"force_topk": st["topk"]}
        gdino_ref.F = proxy
        try:
            pl, pb = gdino_ref.detector_forward(sd_, oc, x[None], text, sm, pid, stages=pst)
        finally:
            gdino_ref.F = RealF
        return ((pb[0] - ref_boxes[0]).abs().max(-1)[0],
                (pl[0] - ref_logits[0]).abs().max(-1)[0] / ref_logits.abs().max())

    sd16 = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in sd.items()}
    eb, el = probe(sd, _F16Operands())              # the yardstick: f16 operands everywhere
    wb, wl = probe(sd16, RealF)                     # second probe: f16 weights only
    for name, mine, emul in (("box", d, eb), ("logit", dl, el)):
        for qt in (0.5, 0.75, 0.9, 0.99, 1.0):
            hq, eq = mine.quantile(qt).item(), emul.quantile(qt).item()
            print(f"{name} err q{qt}: HIP {hq:.2e}  emulated-f16 oracle {eq:.2e}")
            assert hq <= 2.0 * eq + 1e-3, (name, qt, hq, eq)
    sens_b, sens_l = torch.maximum(eb, wb), torch.maximum(el, wl)
    bad = (d > 2e-3 + 20 * sens_b) | (dl > 5e-3 + 20 * sens_l)
    print(f"per-query bound violated by {int(bad.sum())} of {bad.numel()} queries")
    assert bad.float().mean().item() <= 0.02


@torch.no_grad()
def test_detect_threshold_path_matches_oracle_postprocess(dev, small_dino):
    """Row D17: eng.detect() (sigmoid, max over tokens, box_threshold, GD/util/inference.py:70-75) against the
    oracle's postprocess_detections on the oracle's fp32 logits.  With random weights every score saturates at 1.0
    (|logit| ~ 40), so this weight set scales the decoder's final LayerNorm by 0.05 (scores then spread over
    0.67..0.89) and the threshold is the oracle's median score: half of the queries on either side of it.  The
    ill-conditioned tail of random-weight queries (test_detector_stages_match_oracle) can cross any threshold, so the
    kept SET has to agree on >= 90 % of the queries and may not flip more queries, or from farther away, than twice what
    the fp32 oracle itself does when its operands are rounded to f16; the kept boxes / scores agree as in the decoder
    test."""
    from oracle import gdino_ref
    from inklayer_amd import gdino
    sd, oc, _, text = small_dino
    sd = dict(sd)
    for leaf in ("weight", "bias"):
        sd[f"transformer.decoder.norm.{leaf}"] = sd[f"transformer.decoder.norm.{leaf}"] * 0.05
    eng = gdino.GDinoEngine(sd, gdino.GDinoConfig(enc_layers=2, dec_layers=2, num_queries=300), dev, encoded_text=text)
    rs = np.random.RandomState(12)
    img = rs.randint(0, 256, size=(224, 288, 3)).astype(np.uint8)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    x = ((torch.from_numpy(img).permute(2, 0, 1).float() / 255.0) - mean.view(3, 1, 1)) / std.view(3, 1, 1)
    sm, pid = gdino_ref.text_masks_and_position_ids(list(gdino.DEFAULT_TOKEN_IDS))
    dimg = torch.from_numpy(img).to(dev)
    # The two-stage selection RANKS 22k encoder tokens by logits that differ by ~1e-3 relative between f16 and fp32:
    # the selected SET agrees (>= 97 %, test_detector_stages_match_oracle) but its ORDER - which pairs each token with
    # a learned tgt_embed row - does not (64 % of the positions here), and torch.topk's own order is just as arbitrary.
    # Per-query comparisons therefore pin the oracle to the selection the HIP path made.
    st = {}
    lg, bx = eng._forward_eager([dimg], stages=st)
    ref_logits, ref_boxes = gdino_ref.detector_forward(sd, oc, x[None], text, sm, pid, stages={"force_topk": st["topk"].cpu()})
    score = ref_logits[0].sigmoid().max(-1)[0]
    thr = float(score.median())
    assert 0.3 < thr < 0.95 and (score > thr).sum() > 100 and (score <= thr).sum() > 100
    want_xyxy, want_sc = gdino_ref.postprocess_detections(ref_logits[0], ref_boxes[0], thr)
    keep_ref = score > thr
    eng.cfg.box_threshold = thr
    boxes, scores = eng.detect([dimg])[0]
    my_score = lg[0].cpu().sigmoid().max(-1)[0]
    keep_hip = my_score > thr
    # detect() == threshold applied to forward()'s own outputs (the D17 code path itself); the scores to 1 ulp (torch's
    # CPU sigmoid takes a vectorised or a scalar path depending on the tensor's strides)
    assert torch.equal(boxes, bx[0].cpu()[keep_hip]) and torch.allclose(scores, my_score[keep_hip], atol=1e-6, rtol=0)
    agree = (keep_hip == keep_ref).float().mean().item()
    decisive = (score - thr).abs() > 0.05
    print(f"threshold {thr:.4f}: oracle keeps {int(keep_ref.sum())}, HIP keeps {int(keep_hip.sum())}, agreement {agree:.3f}, "
          f"decisive queries {int(decisive.sum())}")
    flips = (keep_hip != keep_ref)
    # yardstick: the fp32 oracle itself with f16-rounded operands (the stated arithmetic) - how many queries cross the
    # threshold there, and from how far away
    import torch.nn.functional as RealF

    class _F16Operands:
        def __getattr__(self, k):
            return getattr(RealF, k)

        def linear(self, a, w, b=None):
            return RealF.linear(a.half().float(), w.half().float(), b)

        def conv2d(self, a, w, b=None, **kw):
            return RealF.conv2d(a.half().float(), w.half().float(), b, **kw)

    gdino_ref.F = _F16Operands()
    try:
        el, _ = gdino_ref.detector_forward(sd, oc, x[None], text, sm, pid, stages={"force_topk": st["topk"].cpu()})
    finally:
        gdino_ref.F = RealF
    eflips = (el[0].sigmoid().max(-1)[0] > thr) != keep_ref
    far = lambda f: (score - thr).abs()[f].max().item() if f.any() else 0.0
    print(f"flipped queries: HIP {int(flips.sum())} (farthest {far(flips):.4f} from the threshold), emulated-f16 oracle "
          f"{int(eflips.sum())} (farthest {far(eflips):.4f})")
    assert agree >= 0.9 and int(flips.sum()) <= 2 * int(eflips.sum()) + 3 and far(flips) <= 2 * far(eflips) + 0.02
    both = keep_hip & keep_ref
    se = (my_score[both] - score[both]).abs()
    b = bx[0].cpu()[both].double().numpy()
    got_xyxy = np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], -1)
    rb = ref_boxes[0][both].double().numpy()
    ref_xyxy = np.stack([rb[:, 0] - rb[:, 2] / 2, rb[:, 1] - rb[:, 3] / 2, rb[:, 0] + rb[:, 2] / 2, rb[:, 1] + rb[:, 3] / 2], -1)
    e = np.abs(got_xyxy - ref_xyxy).max(-1)
    print("kept-in-both: score err p50/p90/max", se.median().item(), se.quantile(0.9).item(), se.max().item(),
          " xyxy err p50/p90/max", np.median(e), np.quantile(e, 0.9), e.max())
    assert se.median().item() < 2e-3 and se.quantile(0.9).item() < 3e-2
    assert np.median(e) < 2e-3 and np.quantile(e, 0.9) < 2e-2 and e.max() < 0.15
    # postprocess_detections' own output format for the common queries (cxcywh -> xyxy in float64)
    idx_ref = torch.nonzero(keep_ref)[:, 0]
    pos = {int(q): i for i, q in enumerate(idx_ref.tolist())}
    rows = [pos[int(q)] for q in torch.nonzero(both)[:, 0].tolist()]
    assert np.allclose(want_xyxy[rows], ref_xyxy) and np.allclose(want_sc[rows], score[both].numpy())


@torch.no_grad()
def test_detector_batch2_and_detect_api(dev, small_dino):
    sd, oc, eng, text = small_dino
    rs = np.random.RandomState(5)
    a = torch.from_numpy(rs.randint(0, 256, size=(224, 224, 3)).astype(np.uint8)).to(dev)
    b = torch.from_numpy(rs.randint(0, 256, size=(224, 224, 3)).astype(np.uint8)).to(dev)
    l2_, b2_ = eng.forward([a, b])
    la, ba = eng.forward([a])
    lb, bb = eng.forward([b])
    assert (b2_[0] - ba[0]).abs().max().item() < 1e-5 and (b2_[1] - bb[0]).abs().max().item() < 1e-5
    assert (l2_[1] - lb[0]).abs().max().item() < 1e-4
    res = eng.detect([a, b], top_n=16)
    assert len(res) == 2 and res[0][0].shape == (16, 4) and res[0][1].shape == (16,)
    res = eng.detect([a])
    assert res[0][0].shape[1] == 4 and (res[0][1] > 0.2).all()


@torch.no_grad()
def test_detector_graph_replay_equals_eager(dev, small_dino):
    """Batch-1 forwards go through a captured HIP graph (latency mode): the replay must reproduce the eager
    forward bit for bit, for new pixels in the static input buffer, and a new input size gets its own graph."""
    sd, oc, eng, text = small_dino
    rs = np.random.RandomState(9)
    imgs = [torch.from_numpy(rs.randint(0, 256, size=(224, 256, 3)).astype(np.uint8)).to(dev) for _ in range(3)]
    other = torch.from_numpy(rs.randint(0, 256, size=(256, 224, 3)).astype(np.uint8)).to(dev)
    assert eng.graph_max_batch >= 1
    eager = [tuple(t.clone() for t in eng._forward_eager([im])) for im in imgs + [other]]
    eng._graphs.clear()
    eng._seen.clear()
    # first sight of a size runs eager, the second captures, then replays; a second size gets its own graph
    for im, (el, eb) in zip(imgs + [other, other] + imgs, eager + [eager[3]] + eager[:3]):
        gl, gb = eng.forward([im])
        assert torch.equal(gl, el) and torch.equal(gb, eb)
    assert len(eng._graphs) == 2
    eng.graph_max_batch = 0                                                 # and it can be switched off
    try:
        gl, gb = eng.forward([imgs[0]])
        assert torch.equal(gl, eager[0][0])
    finally:
        eng.graph_max_batch = 1


@torch.no_grad()
def test_detector_caches_are_bounded(dev, small_dino):
    """A run over sketches of many aspect ratios (one batch-1 forward per size, each size seen twice so that graphs ARE
    captured) keeps at most graph_cache_size graphs / plan_cache_size plans and HBM plateaus instead of growing."""
    sd, oc, eng, text = small_dino
    rs = np.random.RandomState(4)
    eng._graphs.clear(); eng._seen.clear(); eng._plans.clear()
    sizes = [(160 + 32 * i, 224) for i in range(eng.graph_cache_size + 6)]
    mem = []
    for h, w in sizes:
        im = torch.from_numpy(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).to(dev)
        for _ in range(2):
            eng.forward([im])
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated(dev))
    assert len(eng._graphs) == eng.graph_cache_size and len(eng._plans) <= eng.plan_cache_size
    k = eng.graph_cache_size
    growth_early = mem[k - 1] - mem[0]
    growth_late = mem[-1] - mem[k + 1]
    print("allocated MB per size:", [round(m / 2 ** 20, 1) for m in mem])
    assert growth_late < 0.5 * max(growth_early, 1 << 20) + (64 << 20)      # sizes differ a little; no linear growth
    eng._graphs.clear(); eng._seen.clear()


@torch.no_grad()
@pytest.mark.parametrize("T", [4, 3])
def test_folded_fusion_layer_equals_the_full_one(dev, T):
    """ops.fusion_fold (csrc/fusion_fold.hip: the caption's tokens folded through BiAttentionBlock, no per-token
    256 -> 2048 projection) against the layer computed the reference's way in float64 (fuse_modules.py:146-295): image
    update in place, text-side attention output."""
    from inklayer_amd import ops
    g = torch.Generator().manual_seed(11 + T)
    B, S, E, D = 2, 1337, 1024, 256
    v = torch.randn(B * S, D, generator=g) * 1.3 + 0.2
    kl = torch.randn(B * T, 2 * E, generator=g) * 0.7                    # [l_proj | values_l_proj] of LN_l(l)
    Wqv = (torch.randn(2 * E, D, generator=g) / D ** 0.5).half()
    bqv = 0.1 * torch.randn(2 * E, generator=g)
    Wo = (torch.randn(D, E, generator=g) / E ** 0.5).half()
    bo = 0.1 * torch.randn(D, generator=g)
    lng, lnb = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    gam = 0.2 + 0.1 * torch.randn(D, generator=g)
    scale = 256 ** -0.5
    vd = v.to(dev).clone()
    out_l = ops.fusion_fold(vd, B, S, lng.to(dev), lnb.to(dev), 1e-5, kl.to(dev), T, Wqv.to(dev), bqv.to(dev), Wo.to(dev),
                            bo.to(dev), gam.to(dev), scale)
    # the reference's order of operations, float64
    vn = torch.nn.functional.layer_norm(v.double(), (D,), lng.double(), lnb.double(), 1e-5).view(B, S, D)
    q = (vn @ Wqv[:E].double().t() + bqv[:E].double()) * scale
    vv = vn @ Wqv[E:].double().t() + bqv[E:].double()
    k, vl = kl[:, :E].double().view(B, T, E), kl[:, E:].double().view(B, T, E)
    sp = lambda t: t.view(B, -1, 4, 256).transpose(1, 2)                  # [B, 4, n, 256]
    aw = sp(q) @ sp(k).transpose(-1, -2)                                  # [B, 4, S, T]
    ov = (aw.softmax(-1) @ sp(vl)).transpose(1, 2).reshape(B, S, E)
    ol = (aw.transpose(-1, -2).softmax(-1) @ sp(vv)).transpose(1, 2).reshape(B * T, E)
    want_v = (vn + gam.double() * (ov @ Wo.double().t() + bo.double())).view(B * S, D)
    ev = ((vd.double().cpu() - want_v).abs().max() / want_v.abs().max()).item()
    el = ((out_l.double().cpu() - ol).abs().max() / ol.abs().max()).item()
    print(f"T={T}: image update max-rel {ev:.2e}, text-side output max-rel {el:.2e} (f16 store)")
    assert ev < 2e-5 and el < 1e-3
