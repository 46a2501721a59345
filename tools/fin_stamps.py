#!/usr/bin/env python3
"""Phase timeline of the Block-closing ('+fin') convolution launches of the bench model's plan, from the in-kernel stamps of ONE launch each
(fc_debug_set_stamp_op: fc_unet_profile_ops runs the chosen plan entry once more with the stamps on; the stamped launch is the all-in-one
lean flavour of the kernel the plan runs, plus the stamps).

    python tools/fin_stamps.py [substring of the module name ...]        (default: one launch per resolution)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flocoder_amd import _binding as B  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev)
Bn = bench.BATCH
model.reserve(Bn, 32, 32, dev)
rows = model.profile_ops(Bn, repeats=3)
want = sys.argv[1:] or ["downs.0.0", "downs.1.0", "downs.2.0", "downs.3.0", "mid_block1", "ups.3.0", "final_res_block"]
NAMES = {0: "start", 4: "barrier0", 5: "mfma_done", 6: "kreduce", 7: "sums", 14: "stats_out", 12: "group_stats", 13: "applied", 9: "image", 10: "image_bar", 8: "end"}
ORDER = [0, 4, 5, 6, 7, 14, 12, 13, 9, 10, 8]
for i, r in enumerate(rows):
    if not r["kernel"].endswith("+fin") or not any(w in r["module"] for w in want):
        continue
    buf = torch.zeros(8192 * 8 * 16, dtype=torch.int64, device=dev)
    B.check(B.lib().fc_debug_set_stamp_op(i, buf.data_ptr()))
    model.profile_ops(Bn, repeats=1)
    B.check(B.lib().fc_debug_set_stamp_op(-1, None))
    torch.cuda.synchronize()
    st = buf.view(8192, 8, 16).cpu()
    nb = int((st[:, 0, 0] != 0).sum())
    if nb == 0:
        print(f"{r['module']:18s} {r['kernel']:30s} no stamps (not a stamped kernel)")
        continue
    st = st[:nb].double()
    # the waves that run the whole tail: every accumulator wave of an M-split tile, wave 0 of a K-split one (the others hand their partial sums over)
    ksplit = "K2" in r["kernel"] or "K4" in r["kernel"]
    cons = st[:, 0:1] if ksplit else st[:, 0:4]
    load = st[:, 4:8]
    t = lambda k: (cons[:, :, k] - cons[:, :, 0]).median().item() / 2.4e3
    d = lambda a, b: (cons[:, :, a] - cons[:, :, b]).median().item() / 2.4e3
    span = (st[:, :, 8].max(dim=1).values - st[:, :, 0].min(dim=1).values)
    rt = st[:, :, 15].min(dim=1).values * 10.0
    ends = rt + span / 2.4
    print(f"{r['module']:18s} {r['kernel']:30s} {1e3 * r['ms']:6.1f} us timed, {nb} workgroups; first start -> last end {float(ends.max() - rt.min()) / 1e3:.2f} us; "
          f"starts spread over {float(rt.max() - rt.min()) / 1e3:.2f} us")
    print(f"    tail-running wave, us since its start (median): weights requested {t(1):.2f}, stage 0 ready {t(4):.2f}, mfma done {t(5):.2f}, end {t(8):.2f};  "
          f"staging wave: window requested {(load[:, :, 1] - load[:, :, 0]).median().item() / 2.4e3:.2f}, producer's statistics combined {(load[:, :, 2] - load[:, :, 0]).median().item() / 2.4e3:.2f}, "
          f"past the barrier {(load[:, :, 4] - load[:, :, 0]).median().item() / 2.4e3:.2f}, affine tables in LDS {(load[:, :, 5] - load[:, :, 0]).median().item() / 2.4e3:.2f}, "
          f"first window in LDS {(load[:, :, 3] - load[:, :, 0]).median().item() / 2.4e3:.2f}")
    print(f"    tail: K-reduce {d(6, 5):.2f} | block sums {d(7, 6):.2f} | publish / local table {d(14, 7):.2f} | wait for the group's statistics {d(12, 14):.2f} | "
          f"normalise + SiLU + residual {d(2, 12):.2f} | GN(1) partials {d(13, 2):.2f} | to the image {d(3, 13):.2f} | LDS image {d(9, 3):.2f} | barrier {d(10, 9):.2f} | stores {d(8, 10):.2f}"
          f"  = {d(8, 5):.2f} us after the last MFMA")
