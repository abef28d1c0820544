"""`python -m sugar_amd.launch script.py args...`: the script sees the HIP drop-in packages, the stand-ins and its own argv."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_runs_a_script_on_the_drop_ins(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(
        "import sys, json\n"
        "import diff_gaussian_rasterization as dgr, simple_knn, pytorch3d, plyfile\n"
        "from pytorch3d.ops import knn_points\n"
        "print(json.dumps({'argv': sys.argv[1:], 'dgr': dgr.__file__, 'knn': simple_knn.__file__, 'p3d': pytorch3d.__file__,\n"
        "                  'ply': plyfile.__file__, 'name': __name__}))\n")
    env = dict(os.environ, PYTHONPATH="")
    out = subprocess.run([sys.executable, "-m", "sugar_amd.launch", "--quiet", str(script), "-s", "scene", "--flag"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["argv"] == ["-s", "scene", "--flag"] and d["name"] == "__main__"
    assert d["dgr"].startswith(ROOT) and d["knn"].startswith(ROOT)
    # the stand-ins are used only where the real packages are absent (this image: both absent)
    for key in ("p3d", "ply"):
        assert os.path.exists(d[key])


def test_launch_binds_the_reference_modules_when_they_are_importable(tmp_path):
    """with the reference's tree as the script's directory, the opt-in bindings find `sugar_scene.sugar_model`, `sugar_utils.loss_utils`
    and `sugar_scene.sugar_optimizer` (a probe script stands in for train.py: open3d is absent)"""
    import pytest
    from tests import ref_env
    ref = ref_env.reference_root()
    if ref is None:
        pytest.skip("no reference tree")
    probe = tmp_path / "probe.py"
    probe.write_text("print('probe ran')\n")
    code = (
        "import sys, json, types; sys.path.insert(0, %r)\n"
        "sys.modules.setdefault('open3d', types.ModuleType('open3d'))\n"
        "from sugar_amd import launch\n"
        "import os\n"
        "script = os.path.join(%r, 'train.py')\n"
        "done = launch.prepare(script)\n"
        "import sugar_scene.sugar_model as sm, sugar_utils.loss_utils as lu, sugar_scene.sugar_optimizer as so\n"
        "print(json.dumps({'done': {k: (v if not isinstance(v, bool) else int(v)) for k, v in done.items()},\n"
        "  'patched': sorted(sm.SuGaR.__dict__.get('_sugar_amd_original', {})),\n"
        "  'gathers': sorted(sm.SuGaR.__dict__.get('_sugar_amd_row_gather_original', {})),\n"
        "  'ssim': hasattr(lu.ssim, '_sugar_amd_original'), 'opt': hasattr(so.SuGaROptimizer.__dict__['__init__'], '_sugar_amd_original'),\n"
        "  'model_file': sm.__file__}))\n" % (ROOT, ref))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=""), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["model_file"].startswith(ref)
    assert "get_field_values" in d["patched"] and "points" in d["gathers"] and d["ssim"] and d["opt"]
    assert d["done"]["patch_losses"] >= 1 and d["done"]["patch_optimizer"] >= 1
