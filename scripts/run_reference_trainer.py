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
    ap.add_argument("--patch-gathers", action="store_true", help="also give SuGaR's per-Gaussian tensors a HIP row-gather backward")
    ap.add_argument("--trainer", default="coarse_sdf", choices=["coarse_sdf", "coarse_density"])
    ap.add_argument("--patch-optimizer", action="store_true", help="also let SuGaROptimizer's torch.optim.Adam step on the HIP Adam")
    ap.add_argument("--patch-densifier", action="store_true", help="also the densification statistics without boolean-mask indexing (shims.install_densifier)")
    ap.add_argument("--extract", action="store_true",
                    help="then run the reference's coarse-mesh extractor (sugar_extractors/coarse_mesh.py, untouched) up to its "
                         "Poisson step: on the trained model if the training ran to 15000, else on the 3DGS checkpoint")
    ap.add_argument("--skip-training", action="store_true")
    ap.add_argument("--sfm-points", type=int, default=20_000, help="points3D.txt entries of the COLMAP-layout scene (= initial Gaussians)")
    ap.add_argument("--vanilla-cli", type=int, default=0, metavar="N",
                    help="also run `python -m sugar_amd.launch <reference>/gaussian_splatting/train.py ... --iterations N` on a COLMAP-layout "
                         "copy of the scene (the vanilla 3DGS trainer as a command line, densification included)")
    ap.add_argument("--profile-window", type=int, nargs=2, default=None, metavar=("FROM", "TO"),
                    help="torch profiler over these trainer iterations; the kernel table goes to <out>/profile_<tag>.txt")
    ap.add_argument("--refine", type=int, default=0, metavar="N",
                    help="then run the reference's refinement trainer (sugar_trainers/refine.py, untouched) for N iterations on a mesh "
                         "of the scene's surface, six Gaussians per triangle")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "reference_trainer"))
    a = ap.parse_args()
    from oracle import reference_trainer as rt
    work = tempfile.mkdtemp(prefix="sugar_scene_")
    try:
        data = rt.write_dataset(work, P=a.gaussians, n_cams=a.cameras, W=a.width, H=a.height)
        os.makedirs(a.out, exist_ok=True)
        tag = ("" if a.trainer == "coarse_sdf" else a.trainer + "_") + ("dropins_only" if a.no_patch else "patched") + ("_losses" if a.patch_losses else "") + ("_adam" if a.patch_optimizer else "") + ("_gathers" if a.patch_gathers else "") + ("_densifier" if a.patch_densifier else "")
        res = {"finished": False, "model_path": None}
        if not a.skip_training:
          res = rt.run(data, os.path.join(work, "out"), stop_at=a.stop_at, patch_sugar=not a.no_patch, patch_losses=a.patch_losses, patch_optimizer=a.patch_optimizer, trainer=a.trainer, profile_window=a.profile_window, patch_gathers=a.patch_gathers, patch_densifier=a.patch_densifier,
                     log_path=os.path.join(a.out, f"trainer_console_{tag}.log"))
        table = res.pop("profile_table", None)
        if table:
            with open(os.path.join(a.out, f"profile_{tag}.txt"), "w") as f:
                f.write(f"torch profiler, trainer iterations {a.profile_window[0]} .. {a.profile_window[1] - 1}\n" + table + "\n")
        res.update(gaussians=a.gaussians, cameras=a.cameras, width=a.width, height=a.height)
        if a.extract:
            import torch
            ext = rt.run_extractor(data, os.path.join(work, "extract"), res["model_path"] if res["finished"] else None,
                                   patch_sugar=not a.no_patch, patch_gathers=a.patch_gathers,
                                   log_path=os.path.join(a.out, f"extractor_console_{tag}.log"))
            levels = {}
            for lvl, o in (ext.pop("outputs") or {}).items():
                p = o["points"]
                d = p / p.norm(dim=1, keepdim=True).clamp_min(1e-9)
                err = (p.norm(dim=1) - rt.surface_radius(d)).abs()
                levels[str(lvl)] = {"points": int(p.shape[0]), "finite": bool(torch.isfinite(p).all() and torch.isfinite(o["normals"]).all()),
                                    "distance_to_true_surface_median": float(err.median()) if len(err) else None,
                                    "distance_to_true_surface_p90": float(err.quantile(0.9)) if len(err) else None,
                                    "normal_dot_radial_median": float((torch.nn.functional.normalize(o["normals"], dim=1) * d).sum(1).abs().median()) if len(err) else None}
            ext["levels"] = levels
            res["extractor"] = ext
        if a.refine:
            res["refine"] = rt.run_refine(data, os.path.join(work, "refine"), iterations=a.refine, patch_sugar=not a.no_patch,
                                          patch_losses=a.patch_losses, patch_optimizer=a.patch_optimizer, patch_gathers=a.patch_gathers,
                                          log_path=os.path.join(a.out, f"refine_console_{tag}.log"))
        if a.vanilla_cli:
            cdir = rt.write_colmap_dataset(os.path.join(work, "colmap"), P=a.gaussians, n_cams=a.cameras, W=a.width, H=a.height,
                                           n_sfm_points=a.sfm_points)
            flags = [f for f, off in (("--no-patch-losses", not a.patch_losses), ("--no-patch-optimizer", not a.patch_optimizer)) if off]
            v = rt.run_vanilla_cli(cdir, os.path.join(work, "vanilla_out"), iterations=a.vanilla_cli, launcher_flags=flags)
            with open(os.path.join(a.out, f"vanilla_cli_{tag}.log"), "w") as f:
                f.write(" ".join(v["cmd"]) + "\n\n" + v.pop("text"))
            if v["ply"]:
                from sugar_amd import io as sio
                v["gaussians_saved"] = int(sio.load_gaussian_ply(v["ply"])["xyz"].shape[0])
            res["vanilla_cli"] = v
        with open(os.path.join(a.out, f"summary_{tag}.json"), "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
