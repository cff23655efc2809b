"""The N>1 path with REAL processes on the one GPU of the test box: two ranks (gloo as the control plane, both on cuda:0)
render the two tile-row bands of a frame and store their flattened bands into rank 0's frame through a peer-mapped IPC pointer
(`bench.py --gather p2p`, the direct-store variant of SURVEY.md section 8e); rank 0 then holds the whole present payload.
RCCL itself refuses two ranks on one device, so the RCCL gather is covered by the gloo CPU test (tests/test_multigpu_gloo.py)
and `bench.py --fake-world`; a multi-GPU node is the driver's to run."""
import json
import os
import socket
import sys

import numpy as np
import pytest

import spawner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_store_their_bands_into_rank0s_frame(tmp_path):
    out = tmp_path / "frame.npy"
    env = dict(os.environ, SWR_BENCH_ONE_DEVICE="1", SWR_BENCH_DUMP_FRAME=str(out), HSA_ENABLE_IPC_MODE_LEGACY="0")
    # started the way the driver starts it -- `python bench.py --gpus 2 ...`, NOT under torchrun: bench.py launches its own ranks
    # as fresh child processes (launch_ranks_if_needed) and forwards rank 0's JSON line and their exit code
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg3_small", "--steps", "3",
           "--warmup", "1", "--prime", "2", "--gather", "p2p", "--backend", "gloo", "--no-cpu-baseline"]
    r = spawner.run(cmd, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["multi_gpu"]["gather_payload"] == "p2p" and line["multi_gpu"]["frames_resent_after_replay"] == 0
    assert line["config"]["gather_payload"] == "p2p" and line["config"]["gather_bytes_per_pixel"] == 12
    # counters cover every timed frame although the band buffers alternate (BindFramebuffer per frame): 3 steps of the whole frame
    assert line["fragments_tested_per_frame"] > 0 and line["value"] > 0
    # the frame rank 0 ended up with == the Vector4 -> Vector3 flatten of the single-GPU frame
    from softwarerenderer_amd import Device, scenes
    got = np.load(out)
    scene = scenes.cfg3(1024, 1024, (4, 4), (64, 32), tex_size=512)
    dev = Device(0)
    rr = scenes.SceneRenderer(dev, scene)
    dev.reset_stats()
    c, _ = rr.render()
    whole = dev.stats()["fragments_tested"]
    rr.close(); dev.close()
    assert line["fragments_tested_per_frame"] == whole     # summed over both ranks and divided by the timed steps: nothing lost at a Bind
    assert got.shape == (1024, 1024, 3)
    assert np.array_equal(got.view(np.uint32), c[..., :3].view(np.uint32))


@pytest.mark.parametrize("gather", ["rgb32f", "rgba32f"])
def test_the_rccl_leg_runs_end_to_end_with_one_rank(tmp_path, gather):
    """`bench.py --force-dist`: the N>1 branch of bench.py with a world of ONE rank on the nccl (= RCCL) backend, in a fresh child
    process -- init_process_group("nccl", device_id), SetBand, two bound band buffers, flatten, async dist.gather + work.wait(), the
    all-reduced replay flag on the GPU, checkpoints.  What an 8-GPU SCALE run executes first is executed here; rank 0's frame must
    be the single-GPU frame."""
    out = tmp_path / "frame.npy"
    env = dict(os.environ, SWR_BENCH_DUMP_FRAME=str(out), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--config", "cfg3_small", "--steps", "5",
           "--warmup", "2", "--prime", "3", "--gather", gather, "--no-cpu-baseline"]
    r = spawner.run(cmd, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["multi_gpu"]["forced_one_rank_rehearsal"] is True
    assert line["multi_gpu"]["gather_payload"] == gather and line["multi_gpu"]["frames_resent_after_replay"] == 0
    assert line["config"]["gather_bytes_per_pixel"] == (12 if gather == "rgb32f" else 16)
    assert "nccl" not in r.stderr.lower() or "error" not in r.stderr.lower(), r.stderr[-2000:]
    from softwarerenderer_amd import Device, scenes
    got = np.load(out)
    scene = scenes.cfg3(1024, 1024, (4, 4), (64, 32), tex_size=512)
    dev = Device(0)
    rr = scenes.SceneRenderer(dev, scene)
    dev.reset_stats()
    c, _ = rr.render()
    whole = dev.stats()["fragments_tested"]
    rr.close(); dev.close()
    assert line["fragments_tested_per_frame"] == whole
    chan = 3 if gather == "rgb32f" else 4
    assert got.shape == (1024, 1024, chan)
    assert np.array_equal(got.view(np.uint32), c[..., :chan].view(np.uint32))
