"""GPU: bench.py's output contract on a reduced workload (the driver parses this line), and a one-rank rehearsal of the
RCCL merge path (all-reduce on the library's own grid memory)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--grid", "128", "--voxel", "0.02", "--width", "270", "--height", "480", "--steps", "2", "--warmup", "1",
         "--frames-per-step", "4", "--resident-frames", "4", "--cpu-seconds", "0.5", "--cpu-frames", "2"]


def _run(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, text=True,
                         cwd=ROOT, env=e, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract():
    d = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "frames/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "vector-instruction issue" in r["limited_by"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    assert d["config"]["depth_format"] == "f32"
    # (per-launch averages are truncated to integers in the JSON: equal up to that)
    fps = r["frames_per_sweep"]                         # a launch updates a batch of up to 32 frames (here: the 4 frames of a step)
    assert 1.0 <= fps <= 32.0 and r["kernel"] == "tsdf_update_pairs_kernel"
    assert abs(r["bytes_per_launch"] - (8 * r["records_per_launch"] + 8 * r["free_space_bricks_counted_per_launch"] + fps * 4 * 270 * 480)) <= 64
    assert abs(r["us_per_frame"] - 1e3 * r["ms_per_launch"] / fps) < 0.05
    if fps > 1.01:
        one = r["single_frame_per_sweep"]               # the same frames, one per launch
        assert one["frames_per_sweep"] == 1.0 and one["records_per_launch"] < r["records_per_launch"] < fps * one["records_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 0 and abs(d["ms_per_step"] - 1e3 * 4 / d["value"]) / d["ms_per_step"] < 0.05
    # the rest of the path, measured in the same run (registration in the loop, centroid channel, back-projection, extraction)
    rows = d["rows"]
    for k in ("tsdf_plus_centroid_s2_fps", "icp_in_loop_tsdf_fps", "icp_in_loop_tsdf_plus_centroid_fps", "icp_single_chain_us_per_iteration",
              "icp_batch_pairs_per_s", "icp_batch_two_level_from_identity_pairs_per_s", "icp_batch_one_pair_us_per_iteration",
              "backproject_s1_device_us", "backproject_s1_GBps", "extract_centroid_to_host_ms", "outlier_filter_k20_ms"):
        assert rows[k] > 0, k
    assert rows["extract_points"] > 100 and rows["outlier_filter_kept"] <= rows["extract_points"]
    # every row is a median of 5 samples, its [min, max] beside it; the centroid channel has a row of its own with COUNTED bytes
    assert rows["repetitions"] == 5 and all(lo <= rows[k] <= hi for k, (lo, hi) in rows["spread"].items())
    cen = rows["centroid"]
    assert cen["record_updates_per_frame"] > 0 and abs(cen["bytes_per_frame"] - (7 * (270 // 2) * (480 // 2) + 64 * cen["record_updates_per_frame"])) <= 64
    assert abs(cen["frac_of_8TBps"] - cen["achieved_GBps"] / 8000.0) < 1e-3
    # the reference's own CPU path (restated), bounded sample, beside the port
    rr = c["restated_reference_path"]
    assert rr["kind"] == "restated reference" and rr["value"] > 0 and rr["cores"] == 1 and "voxel centroid" in rr["sample"]
    # stream concurrency and runtime versions are recorded
    assert d["config"]["hw_queues"]["effective_queues"] >= 1 and d["config"]["hip"]["hip_runtime"] > 0
    assert d["config"]["invalid_pixel_fraction"] == 0.0


def test_single_rank_rccl_merge_rehearsal():
    d = _run(["--force-dist", "--no-cpu-baseline", "--no-rows", "--centroid", "--depth-format", "u16"], env={"MASTER_PORT": "29541"})
    assert d["value"] > 0 and d["config"]["centroid_channel"] is True and d["config"]["depth_format"] == "u16"
    assert d["config"]["grid_merges_in_timed_region"] == 1 and d["rows"] is None
    r = d["roofline"]
    assert abs(r["bytes_per_launch"] - (8 * r["records_per_launch"] + 8 * r["free_space_bricks_counted_per_launch"] + r["frames_per_sweep"] * 2 * 270 * 480)) <= 64


def test_two_ranks_on_one_gpu_rehearse_the_multi_gpu_bench():
    """`python bench.py --gpus 2` outside torchrun starts its own two ranks (fresh processes, before any GPU call); on this one-GPU
    box they share the device and merge through gloo.  Every rank flies a turn with the SAME frame spacing as the N = 1 run."""
    d = _run(["--gpus", "2", "--no-cpu-baseline", "--no-rows"], env={"TL3D_SHARE_DEVICE": "1", "TL3D_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    c = d["config"]
    assert c["dist_ranks"] == 2 and c["dist_backend"] == "gloo" and c["grid_merges_in_timed_region"] == 1 and len(c["merge_ms"]) == 1
    one = _run(["--no-cpu-baseline", "--no-rows"])
    assert one["config"]["dist_ranks"] == 1 and "90.00 degrees apart on every rank" in one["config"]["workload"]
    assert "90.00 degrees apart on every rank" in c["workload"]
    assert abs(d["ms_per_step"] - 1e3 * 2 * 4 / d["value"]) / d["ms_per_step"] < 0.05
