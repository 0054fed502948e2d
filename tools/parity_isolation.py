#!/usr/bin/env python
"""Where does the full-size eval discrepancy between the product and the CPU oracle come from?

  python tools/parity_isolation.py --tag gwc_gc_384x1248 [--label default] [--out gpurun_out/parity_isolation.jsonl]

Splits GwcNet_GC(192) at the 1/4-resolution features (the boundary between the stock MIOpen 2-D CNN and the hand-written
HIP path) and crosses the two halves with the oracle's:

  full      product features (MIOpen)  -> product 3-D path (HIP)       the shipped model
  A         ORACLE features (CPU)      -> product 3-D path (HIP)       error of the hand-written part alone
  B         product features (MIOpen)  -> ORACLE 3-D path (CPU)        effect of the 2-D CNN's rounding alone
  oracle    oracle features            -> oracle 3-D path              the yard-stick (north_star: 1e-3 max-abs)

Every row is reported against the oracle over ALL pixels and against the reference's own fp64 run (fixture
tests/golden/fullsize_eval.npz, every 4th pixel).  Run it under different MIOpen solver environments
(MIOPEN_DEBUG_CONV_WINOGRAD=0, ...) with --label to attribute row B to a solver family.  Test infrastructure:
imports oracle/, never imported by the product.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="gwc_gc_384x1248")
    ap.add_argument("--label", default="default")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_isolation.jsonl"))
    ap.add_argument("--no-benchmark", action="store_true", help="torch.backends.cudnn.benchmark = False (MIOpen immediate mode)")
    args = ap.parse_args()
    from oracle import torch_oracle as O
    from stereo_toolbox_amd.models import GwcNet_GC
    from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
    torch.backends.cudnn.benchmark = not args.no_benchmark
    H, W = (int(v) for v in args.tag.split("_")[-1].split("x"))
    D = 192
    m = GwcNet_GC(D)
    sd = m.state_dict()
    fill_state_dict(sd)
    m.load_state_dict(sd)
    ref_sd = {k: v.clone() for k, v in sd.items()}
    dev = torch.device("cuda:0")
    m = m.to(dev).eval()
    left, right = synthetic_tensor((1, 3, H, W), 1), synthetic_tensor((1, 3, H, W), 2)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_eval.npz"))
    s = int(gold["stride"])
    ref64 = torch.from_numpy(gold[args.tag + "_64"])

    with torch.no_grad():
        cx = O.Ctx(ref_sd, False)
        ogl, ocl = O.features_gwc(cx, left, True)
        ogr, ocr = O.features_gwc(cx, right, True)
        ref = O.gwcnet_aggregate(cx, ogl, ogr, ocl, ocr, D, H, W)
        from stereo_toolbox_amd.models.features2d import run_pair
        fl, fr = run_pair(m.feature_extraction, left.to(dev), right.to(dev), False)
        full = m.aggregate(fl, fr, H, W).cpu()
        a = m.aggregate({"gwc_feature": ogl.to(dev), "concat_feature": ocl.to(dev)},
                        {"gwc_feature": ogr.to(dev), "concat_feature": ocr.to(dev)}, H, W).cpu()
        pgl, pcl = fl["gwc_feature"].cpu().contiguous(), fl["concat_feature"].cpu().contiguous()
        pgr, pcr = fr["gwc_feature"].cpu().contiguous(), fr["concat_feature"].cpu().contiguous()
        b = O.gwcnet_aggregate(cx, pgl, pgr, pcl, pcr, D, H, W)

    def vs(x):
        return {"max_abs_vs_oracle_all_px": float((x - ref).abs().max()),
                "mean_abs_vs_oracle": float((x - ref).abs().mean()),
                "max_abs_vs_reference_fp64_sampled": float((x[:, ::s, ::s].double() - ref64).abs().max())}

    def fdiff(p, o):
        return {"max_abs": float((p - o).abs().max()), "rel_to_max": float((p - o).abs().max() / o.abs().max()),
                "rms_rel": float(((p - o).double().pow(2).mean().sqrt() / o.double().pow(2).mean().sqrt()))}

    rec = {"tag": args.tag, "label": args.label, "cudnn_benchmark": torch.backends.cudnn.benchmark,
           "env": {k: v for k, v in os.environ.items() if k.startswith("MIOPEN_")},
           "full (MIOpen features -> HIP 3-D)": vs(full),
           "A (oracle features -> HIP 3-D)": vs(a),
           "B (MIOpen features -> oracle 3-D)": vs(b),
           "oracle": {"max_abs_vs_reference_fp64_sampled": float((ref[:, ::s, ::s].double() - ref64).abs().max()),
                      "reference_fp32_vs_fp64_all_px": float(gold[args.tag + "_e32"])},
           "features MIOpen vs oracle": {"gwc_left": fdiff(pgl, ogl), "gwc_right": fdiff(pgr, ogr),
                                         "concat_left": fdiff(pcl, ocl), "concat_right": fdiff(pcr, ocr)}}
    line = json.dumps(rec)
    print(line)
    if os.path.isdir(os.path.dirname(args.out)):
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
