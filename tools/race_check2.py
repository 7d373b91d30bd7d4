"""Where does the second of two in-flight host-to-host steps go wrong?  Re-does submit_host's second step by hand and
keeps its intermediate tensors for comparison with a synchronous reference.  Development aid."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import bench
    from inklayer_amd import pipeline, synthetic
    dev = torch.device("cuda:0")
    det, seg, _ = bench.build_engines(dev, 0, 1, 8)
    imgs = [synthetic.synthetic_sketch(i) for i in range(8)]
    p = pipeline.InkLayerPipeline(det, seg, overlap=True)
    p.encoder_first = False
    host = p.pinned_like(imgs)
    ref_raw = p.upload(imgs)
    ref_det_in, ref_sam_in, sizes = p.preprocess(ref_raw)
    ref = p.run_prepared(ref_det_in, ref_sam_in, sizes, top_n=16)
    torch.cuda.synchronize()
    ref_boxes = [np.asarray(r.boxes_xyxy_norm).copy() for r in ref]
    ref_emb = seg.encode(ref_sam_in, chan_reverse=True).clone()
    ref_lg, ref_bx = det.forward(ref_det_in, allow_graph=False)
    ref_lg, ref_bx = ref_lg.clone(), ref_bx.clone()
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream(dev)
    stash = []
    orig_post = det.postprocess

    def post(host_both, top_n=None):
        stash.append(host_both.clone())
        return orig_post(host_both, top_n=top_n)
    det.postprocess = post
    orig_fwd = det._forward_eager
    st_ref = {}
    orig_fwd(ref_det_in, st_ref)
    torch.cuda.synchronize()
    st_runs = []

    def fwd(images, stages=None):
        d = {}
        out = orig_fwd(images, d)
        st_runs.append(d)
        return out
    det._forward_eager = fwd
    ref_both = torch.cat([ref_lg, ref_bx], -1).cpu()
    for rep in range(30):
        t1 = p.submit_host(host, top_n=16)
        with torch.cuda.stream(p.s_h2d):
            raw2 = [t.to(dev, non_blocking=True) for t in host]
            up = torch.cuda.Event()
            up.record(p.s_h2d)
        cur.wait_event(up)
        for r in raw2:
            r.record_stream(cur)
        det_in2, sam_in2, sizes2 = p.preprocess(raw2)
        res2 = p.run_prepared(det_in2, sam_in2, sizes2, top_n=16)
        boxes2 = [np.asarray(r.boxes_xyxy_norm).copy() for r in res2]
        torch.cuda.synchronize()
        p.collect_host(t1)
        bad_raw = [i for i in range(8) if not torch.equal(raw2[i], ref_raw[i])]
        bad_det = [i for i in range(8) if not torch.equal(det_in2[i], ref_det_in[i])]
        bad_sam = [i for i in range(8) if not torch.equal(sam_in2[i], ref_sam_in[i])]
        bad_box = [i for i in range(8) if not np.array_equal(boxes2[i], ref_boxes[i])]
        # recompute from the kept inputs, synchronously: were the INPUTS the problem or the compute?
        lg, bx = det.forward(det_in2, allow_graph=False)
        torch.cuda.synchronize()
        for k, hb in enumerate(stash):
            d = (hb != ref_both)
            if d.any():
                per_img = d.flatten(1).sum(1).tolist()
                mag = (hb - ref_both).abs()
                print(f"   rep {rep} detector output #{k} ({'first' if k == 0 else 'second'} step): differing elements per image {per_img} "
                      f"of {hb[0].numel()}; max abs diff {mag.max().item():.3e}, median of the differing {mag[d].median().item():.3e}; "
                      f"logits part differs {bool(d[..., :hb.shape[-1] - 4].any())}, boxes part differs {bool(d[..., -4:].any())}", flush=True)
        stash.clear()
        if bad_box:
            for k, d in enumerate(st_runs):
                for name, t in d.items():
                    r = st_ref.get(name)
                    if not torch.is_tensor(t) or not torch.is_tensor(r) or t.shape != r.shape:
                        continue
                    if not torch.equal(t, r):
                        B = 8
                        per = (t.reshape(B, -1) != r.reshape(B, -1)).sum(1).tolist() if t.numel() % B == 0 else "?"
                        print(f"      forward #{k} stage {name:14s} differs; differing elements per image {per}", flush=True)
                        if name == "src":
                            d = (t != r).nonzero()
                            rows = sorted(set(d[:, 0].tolist()))
                            S = t.shape[0] // 8
                            for row in rows[:12]:
                                cols = d[d[:, 0] == row][:, 1].tolist()
                                print(f"         src row {row} = image {row // S} token {row % S} of {S}, cols {cols[0]}..{cols[-1]} ({len(cols)}): "
                                      f"got {t[row, cols[0]:cols[0] + 4].tolist()} want {r[row, cols[0]:cols[0] + 4].tolist()}", flush=True)
        st_runs.clear()
        print(f"rep {rep}: raw differs {bad_raw}  det_in differs {bad_det}  sam_in differs {bad_sam}  boxes differ {bad_box}  "
              f"| detector re-run on the kept inputs equals reference: {torch.equal(lg, ref_lg) and torch.equal(bx, ref_bx)}",
              flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
