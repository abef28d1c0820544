"""How many pixels of a deep tile's 8x8 blocks are still accumulating as the block's walk goes on (BASELINE config 4, one view):
for the deepest blocks, the share of (entry, pixel) evaluations spent on pixels that are still live -- what a one-wave walk wastes
on lanes that have stopped -- and the number of live pixels by list position."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sugar_amd import synthetic as syn
from tests import parity_utils as pu
scene, cams, bg = syn.make_config("config4")[:3]
cam = cams[3]
W, H = cam.image_width, cam.image_height
o = pu.run_hip(scene, cam, bg)
ts = o["tile_start"].astype(np.int64); n = np.diff(ts)
nc = o["n_contrib"].reshape(H, W).astype(np.int64); fT = o["final_T"].reshape(H, W)
gx = (W + 15) // 16
out = {"tiles": int(n.size), "deepest_lists": [int(v) for v in np.sort(n)[-5:]], "blocks": []}
tot_eval = tot_live = 0
for t in np.argsort(n)[-12:]:
    tx, ty = t % gx, t // gx
    for sub in range(4):
        x0, y0 = tx * 16 + 8 * (sub & 1), ty * 16 + 8 * (sub >> 1)
        blk = nc[y0:y0 + 8, x0:x0 + 8].reshape(-1)
        if blk.size == 0: continue
        last = int(blk.max())                      # the block's walk ends behind its deepest contributor at the earliest
        walked = max(last, 1)
        # a pixel is live up to its own last contributor (then it has stopped, or nothing later contributes to it)
        live_at = [(blk >= p).sum() for p in (1, walked // 4, walked // 2, 3 * walked // 4, walked)]
        tot_eval += walked * 64; tot_live += int(blk.sum())
        out["blocks"].append({"tile": int(t), "block": sub, "list": int(n[t]), "deepest_contributor": last,
                              "pixels_with_a_contributor_at_or_behind_0_25_50_75_100pct": [int(v) for v in live_at],
                              "min_final_T": float(fT[y0:y0 + 8, x0:x0 + 8].min())})
out["share_of_pair_evaluations_on_pixels_not_yet_past_their_last_contributor"] = tot_live / max(tot_eval, 1)
print(json.dumps(out, indent=1))
