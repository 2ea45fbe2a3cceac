"""CPU: bench.py's self-launch for N > 1 (`python bench.py --gpus N` from a plain shell must start its own
ranks).  No GPU here: the launcher is exercised with torch.distributed.run over a stand-in rank script (gloo-free: the
stand-in only prints), and the real command line is checked textually."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench touches neither torch nor the GPU)


def test_importing_bench_loads_neither_torch_nor_the_library():
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "assert 'torch' not in sys.modules and 'pathtracer_amd' not in sys.modules; print('clean')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    assert out.strip() == "clean"


def test_launch_command_is_the_drivers_form():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]


def test_main_becomes_the_launcher_before_touching_the_gpu(monkeypatch):
    seen = {}

    def fake(n, argv):
        seen["n"], seen["argv"] = n, list(argv)
        seen["torch_loaded"] = "torch" in sys.modules and hasattr(sys.modules["torch"], "cuda") and \
            sys.modules["torch"].cuda.is_initialized()
        return 7

    monkeypatch.setattr(bench, "self_launch", fake)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and seen["n"] == 2 and seen["argv"] == ["--gpus", "2", "--steps", "3"] and not seen["torch_loaded"]
    # --force-dist on one GPU goes through the same launcher
    monkeypatch.setattr(sys, "argv", ["bench.py", "--force-dist"])
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["n"] == 1


def test_self_launch_runs_ranks_and_relays_one_line(monkeypatch, tmp_path, capfd):
    """Two real ranks under torch.distributed.run (a stand-in script: rank 0 prints the JSON line, both print noise)."""
    script = tmp_path / "rank.py"
    script.write_text("import os, json\n"
                      "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                      "print('noise from rank %d' % r, flush=True)\n"
                      "if r == 0: print(json.dumps({'metric': 'Mrays/s', 'n_gpus': w}), flush=True)\n")
    real = bench.launch_command

    def cmd(n, argv, port):
        c = real(n, argv, port)
        c[c.index(os.path.join(ROOT, "bench.py"))] = str(script)
        return c

    monkeypatch.setattr(bench, "launch_command", cmd)
    rc = bench.self_launch(2, [])
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "Mrays/s", "n_gpus": 2}
    assert "noise from rank 1" in err
    # a failing rank's exit code comes back
    script.write_text("import sys; sys.exit(3)\n")
    assert bench.self_launch(2, []) != 0


def test_lavapipe_probe_reports_what_it_looked_for():
    msg = bench.probe_lavapipe()
    assert msg.startswith(("unavailable:", "present")) and "ICD" in msg


# ---------------------------------------------------------------- per-workload counter constants (round 3) ----
def test_profile_constants_are_keyed_by_workload_and_kernel_sources(monkeypatch, tmp_path, capsys):
    """bench.py prices a run with the committed PMC summary of ITS workload only: the file name carries the scene (and the
    soup's size), the summary records the bench line's profile_workload and the hash of the kernel sources, and a mismatch
    of either makes load_profile_json return None (the line then has no VALU roofline rather than a wrong one)."""
    a = bench.parse(["--scene", "soup", "--tris", "1e6", "--builder", "lbvh"])
    b = bench.parse(["--scene", "soup", "--tris", "1e8", "--builder", "lbvh"])
    c = bench.parse([])
    assert bench.profile_suffix(a) == "_soup_1000000" and bench.profile_suffix(b) == "_soup" and bench.profile_suffix(c) == ""
    assert bench.profile_suffix(bench.parse(["--scene", "teapot"])) == "_teapot"
    assert bench.workload_key(a) != bench.workload_key(b) and "tris1000000" in bench.workload_key(a)
    assert bench.workload_key(c) == "cornell 1920x1080 spp16 depth8 ploc persistent"
    assert bench.workload_key(bench.parse(["--steps", "3"])) == bench.workload_key(c)  # the frame count does not enter
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "abc")
    good = {"kernel_sources_sha": "abc", "workload": bench.workload_key(c), "valu_wave_insts_per_ray": 12.3}
    (tmp_path / "profiles" / "valu_paths.json").write_text(json.dumps(good))
    assert bench.load_profile_json("valu_paths.json", bench.workload_key(c))["valu_wave_insts_per_ray"] == 12.3
    assert bench.load_profile_json("valu_paths.json", bench.workload_key(a)) is None       # another workload
    assert "not used" in capsys.readouterr().err
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "other")
    assert bench.load_profile_json("valu_paths.json", bench.workload_key(c)) is None       # other kernel sources
    assert bench.load_profile_json("missing.json", bench.workload_key(c)) is None


def test_committed_profiles_match_the_kernel_sources_in_the_tree():
    """The summaries under profiles/ that bench.py loads were taken on THESE kernel sources (otherwise the round-end bench
    line silently loses its roofline): regenerate with scripts/gpu_round_profile.sh / gpu_scene_profile.sh + collect_profile.sh."""
    sha = bench.kernel_source_hash()
    for name in ("valu_paths", "traffic_paths", "valu_paths_viking", "traffic_paths_viking", "valu_paths_teapot",
                 "traffic_paths_teapot", "valu_paths_soup", "traffic_paths_soup", "valu_paths_soup_1000000",
                 "traffic_paths_soup_1000000"):
        d = json.load(open(os.path.join(ROOT, "profiles", name + ".json")))
        assert d["kernel_sources_sha"] == sha, name
        assert d.get("workload"), name


def test_cpu_baseline_on_a_pixel_sample(monkeypatch):
    """cpu_baseline_pixels (the scenes other than the Cornell box): the oracle on a seeded pixel sample, rays / second."""
    sys.path.insert(0, ROOT)
    import pathtracer_amd as pa
    a = bench.parse(["--scene", "teapot", "--width", "64", "--height", "48", "--spp-per-frame", "2", "--depth", "3",
                     "--cpu-sample-pixels", "50"])
    sc, view, proj, tmax, _ = bench.build_scene(a, pa)
    a.instances = a.instances[:9]  # nine teapots are enough here
    out = bench.cpu_baseline_pixels(a, sc, view, proj, tmax, budget_s=0.05)
    assert out["kind"] == "port" and out["unit"] == "Mrays/s" and out["value"] > 0 and out["cores"] >= 1
    assert "random pixels of 64x48" in out["sample"] and "no top-level structure" in out["sample"]


def test_bench_arguments_of_round_3():
    """--dist-backend gloo (N ranks rehearsed on fewer GPUs) and --ramp-ms (untimed clock ramp) parse; their defaults are
    the measured path: RCCL, 150 ms."""
    a = bench.parse([])
    assert a.dist_backend == "nccl" and a.ramp_ms == 150.0 and a.scaling == "weak"
    a = bench.parse(["--gpus", "4", "--dist-backend", "gloo", "--ramp-ms", "0", "--scaling", "strong"])
    assert a.gpus == 4 and a.dist_backend == "gloo" and a.ramp_ms == 0.0 and a.scaling == "strong"
