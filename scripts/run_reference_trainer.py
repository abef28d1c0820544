#!/usr/bin/env python
"""Runs the reference's own coarse trainer (sugar_trainers/coarse_sdf.py, untouched) on the HIP drop-ins over a synthetic
on-disk scene and prints one JSON line: the losses it logged, the events it announced, iterations/s before and after the SDF
regularisation starts.  See oracle/reference_trainer.py.  GPU needed.

    python scripts/run_reference_trainer.py --stop-at 9300 --out gpurun_out/r04/reference_trainer
"""
import argparse
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stop-at", type=int, default=9300, help="trainer iteration to stop after (it starts at 7000, ends at 15000)")
    ap.add_argument("--gaussians", type=int, default=60_000)
    ap.add_argument("--cameras", type=int, default=48)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--no-patch", action="store_true", help="drop-in packages only; SuGaR's own tensor code for the field methods")
    ap.add_argument("--patch-losses", action="store_true", help="also bind the trainer's `ssim` to the HIP loss kernels")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "reference_trainer"))
    a = ap.parse_args()
    from oracle import reference_trainer as rt
    work = tempfile.mkdtemp(prefix="sugar_scene_")
    try:
        data = rt.write_dataset(work, P=a.gaussians, n_cams=a.cameras, W=a.width, H=a.height)
        os.makedirs(a.out, exist_ok=True)
        tag = ("dropins_only" if a.no_patch else "patched") + ("_losses" if a.patch_losses else "")
        res = rt.run(data, os.path.join(work, "out"), stop_at=a.stop_at, patch_sugar=not a.no_patch, patch_losses=a.patch_losses,
                     log_path=os.path.join(a.out, f"trainer_console_{tag}.log"))
        res.update(gaussians=a.gaussians, cameras=a.cameras, width=a.width, height=a.height)
        with open(os.path.join(a.out, f"summary_{tag}.json"), "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
