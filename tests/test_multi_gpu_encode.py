"""CPU: the multi-GPU side of the hooked encoder without a node — eight encoder instances, one per GPU ordinal, against the CPU test double
(SVT_HIP_MOCK_DEVICES=8): every instance reports its own ordinal, none falls back, and the eight bitstreams equal eight single runs; two encoder instances inside ONE
process share the hooks' contexts and survive the first one's deinit (the reference count of svt_hip_hooks_enc_init / _deinit); an ordinal that names no device keeps
the C path with a log line; bench.py --gpus 2 under a launcher checks WORLD_SIZE."""
import argparse
import hashlib
import os
import subprocess
import sys

import pytest

import e2e_common as E
from conftest import ROOT

pytestmark = pytest.mark.skipif(not E.have_apps(), reason="oracle/_ref encoders not built (needs /root/reference)")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_eight_instances_one_per_gpu_ordinal(tmp_path):
    import multi_gpu_encode as M
    a = argparse.Namespace(gpus=8, width=176, height=144, frames=4, preset=8, q=40, lp=2, hooks="all", simd=False, numa=True, check=True, app=None, workdir=str(tmp_path), timeout=900)
    out = M.run(a, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_MOCK_DEVICES": "8"})
    assert [i["ordinal"] for i in out["instances"]] == list(range(8)), out
    assert all(i["rc"] == 0 and i["devices"] == 8 and i["by"] == "SVT_HIP_DEVICE" and i["fallbacks"] == 0 and i["mock"] for i in out["instances"]), out
    assert out["identical_to_single_runs"] == [True] * 8, out
    assert out["aggregate_fps_encoder_clock"] > 0 and out["frames"] == 32
    # ... and N concurrent encodes sharing ONE device (--same-device: every instance on ordinal 0)
    a.gpus, a.check, a.same_device = 3, False, True
    out = M.run(a, {"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_MOCK_DEVICES": "8"})
    assert [i["ordinal"] for i in out["instances"]] == [0, 0, 0] and all(i["rc"] == 0 and i["fallbacks"] == 0 for i in out["instances"]), out


def test_ordinal_without_a_device_keeps_the_c_path(tmp_path):
    clip = str(tmp_path / "c.yuv")
    E.make_clip(clip, 176, 144, 3, seed=2)
    ref = E.encode(E.APP_REF, clip, 176, 144, 3, 8, 40, 8, str(tmp_path / "ref"))
    got = E.encode(E.APP_HIP, clip, 176, 144, 3, 8, 40, 8, str(tmp_path / "hip"), env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_DEVICE": "3", "SVT_HIP_MOCK_DEVICES": "2"})
    assert "svt_hip_init failed" in got["log"] and got["ivf"] == ref["ivf"] and got["recon"] == ref["recon"]
    # target_socket (-ss, the reference's CPU-affinity knob) doubles as the ordinal only when it names a device: -ss 1 on a one-GPU host stays on device 0
    got = E.encode(E.APP_HIP, clip, 176, 144, 3, 8, 40, 8, str(tmp_path / "hip2"), env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_MOCK_DEVICES": "1"}, extra_args=["-ss", "1"])
    assert "svt_hip_device ordinal=0 of 1 (default)" in got["log"] and got["ivf"] == ref["ivf"] and got["hooks"]["me"][0] > 0
    got = E.encode(E.APP_HIP, clip, 176, 144, 3, 8, 40, 8, str(tmp_path / "hip3"), env_extra={"LD_LIBRARY_PATH": E.MOCK_DIR, "SVT_HIP_HOOKS": "all", "SVT_HIP_MOCK_DEVICES": "2"}, extra_args=["-ss", "1"])
    assert "svt_hip_device ordinal=1 of 2 (target_socket)" in got["log"] and got["ivf"] == ref["ivf"]


def test_two_encoder_instances_in_one_process(tmp_path):
    """-nch 2: both channels share the hooks' contexts, pool, block cache and resident table; the shorter one deinitialises while the other still encodes"""
    a, b = str(tmp_path / "a.yuv"), str(tmp_path / "b.yuv")
    E.make_clip(a, 352, 288, 4, seed=3)
    E.make_clip(b, 352, 288, 9, seed=9)
    ra = E.encode(E.APP_REF, a, 352, 288, 4, 6, 35, 8, str(tmp_path / "ra"))
    rb = E.encode(E.APP_REF, b, 352, 288, 9, 6, 35, 8, str(tmp_path / "rb"))
    env = dict(os.environ, LD_LIBRARY_PATH=E.MOCK_DIR, SVT_HIP_HOOKS="all", SVT_HIP_RESIDENT="1")
    oa, ob = str(tmp_path / "ma.ivf"), str(tmp_path / "mb.ivf")
    cmd = [E.APP_HIP, "-nch", "2", "-i", a, b, "-w", "352", "352", "-h", "288", "288", "-n", "4", "9", "--preset", "6", "6", "--fps", "30", "30", "-q", "35", "35", "--lp", "8", "8", "-b", oa, ob]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
    assert md5(oa) == ra["ivf"] and md5(ob) == rb["ivf"]
    log = r.stdout + r.stderr
    assert log.count("svt_hip MOCK") >= 1 and "fallback=0" in log and not [l for l in log.splitlines() if l.startswith("svt_hip_hook ") and not l.endswith("fallback=0")]
    # the first instance to end releases every page-locked range (the tables are shared and the instance frees its pictures); the other re-registers its own
    assert "svt_hip_pins released_while_other_instances_ran=1" in log, [l for l in log.splitlines() if l.startswith("svt_hip_")]


def test_bench_under_a_launcher_checks_world_size():
    """bench.py --gpus 2 started as ONE rank of a WORLD_SIZE-1 job must refuse (the driver's launcher sets WORLD_SIZE = N); without a launcher it spawns its ranks
    itself -- which needs GPUs, so only the refusal is exercised here"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
