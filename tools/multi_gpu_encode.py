"""BASELINE configs[4] for the hooked ENCODER: N concurrent streams, one encoder instance per GPU (frames / streams shard one-per-GPU, no collective).

    python tools/multi_gpu_encode.py --gpus 8 --width 3840 --height 2160 --frames 8 --preset 6 [--check]

Instance k runs oracle/_ref/SvtAv1EncApp_hip[_simd] with SVT_HIP_DEVICE=k (the reference restricts its own `-ss` / target_socket to -1..1, EbEncHandle.c:2803, so the GPU
ordinal travels in the environment; `-ss` is set to the instance's half of the node when --numa is given: GPUs 0..N/2-1 hang off socket 0 on the MI355X boxes) on its
own synthetic clip; all instances start together.  Prints one JSON line: per-instance ordinal (as the instance itself reports it: "svt_hip_device ordinal=k of n"),
frames per second by the encoder's own clock, the aggregate, and with --check whether every bitstream equals the one a single instance produces on its own.
On a box without GPUs the same script runs against the CPU test double (LD_LIBRARY_PATH=oracle/_ref/mock, SVT_HIP_MOCK_DEVICES=N): tests/test_multi_gpu_encode.py."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def _cmd(app, clip, a, out, socket=None):
    c = [app, "-i", clip, "-w", str(a.width), "-h", str(a.height), "-n", str(a.frames), "--preset", str(a.preset), "--fps", "30", "-q", str(a.q), "--lp", str(a.lp), "-b", out]
    return c + (["-ss", str(socket)] if socket is not None else [])


def run(a, env_extra=None):
    import e2e_common as E   # make_clip (numpy only)
    app = a.app or os.path.join(ROOT, "oracle", "_ref", "SvtAv1EncApp_hip_simd" if a.simd else "SvtAv1EncApp_hip")
    wd = a.workdir or os.path.join(ROOT, "gpurun_out", "multi_gpu")
    os.makedirs(wd, exist_ok=True)
    base = dict(os.environ)
    base.update(env_extra or {})
    base["SVT_HIP_HOOKS"] = a.hooks
    clips = []
    for k in range(a.gpus):
        clip = os.path.join(wd, f"stream{k}.yuv")
        E.make_clip(clip, a.width, a.height, a.frames, seed=100 + k, bd=8)
        clips.append(clip)
    t0 = time.time()
    procs = []
    for k in range(a.gpus):
        env = dict(base, SVT_HIP_DEVICE=str(0 if getattr(a, "same_device", False) else k))
        if getattr(a, "reference", False): env.pop("SVT_HIP_HOOKS", None)   # the same binary with no hook = the reference encoder
        sock = (0 if k < (a.gpus + 1) // 2 else 1) if a.numa else None
        procs.append(subprocess.Popen(_cmd(app, clips[k], a, os.path.join(wd, f"stream{k}.ivf"), sock), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=a.timeout)[0] for p in procs]
    wall = time.time() - t0
    inst = []
    for k, (p, log) in enumerate(zip(procs, logs)):
        m = re.search(r"svt_hip_device ordinal=(\d+) of (\d+) \((\w+)\)", log)
        enc_ms = re.search(r"Total Encoding Time:\s+(\d+) ms", log)
        fps = re.search(r"Average Speed:\s+([0-9.]+) fps", log)
        fb = sum(int(x) for x in re.findall(r"svt_hip_hook \w+ handled=\d+ fallback=(\d+)", log))
        inst.append({"instance": k, "rc": p.returncode, "ordinal": int(m.group(1)) if m else None, "devices": int(m.group(2)) if m else None, "by": m.group(3) if m else None,
                     "fps": float(fps.group(1)) if fps else None, "encode_ms": int(enc_ms.group(1)) if enc_ms else None, "fallbacks": fb, "mock": "svt_hip MOCK" in log, "md5": _md5(os.path.join(wd, f"stream{k}.ivf")) if p.returncode == 0 else None})
    out = {"gpus": a.gpus, "wall_s": round(wall, 2), "frames": a.frames * a.gpus, "aggregate_fps_wall": round(a.frames * a.gpus / wall, 3),
           "aggregate_fps_encoder_clock": round(sum(i["fps"] or 0 for i in inst), 3), "instances": inst}
    if a.check:   # one instance at a time on device 0: the same clip must code to the same bitstream wherever it runs
        same = []
        for k in range(a.gpus):
            o = os.path.join(wd, f"single{k}.ivf")
            r = subprocess.run(_cmd(app, clips[k], a, o), env=dict(base, SVT_HIP_DEVICE="0"), capture_output=True, text=True, timeout=a.timeout)
            same.append(r.returncode == 0 and _md5(o) == inst[k]["md5"])
        out["identical_to_single_runs"] = same
    for f in os.listdir(wd):
        if f.endswith((".yuv", ".ivf")):
            os.remove(os.path.join(wd, f))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--preset", type=int, default=6)
    ap.add_argument("--q", type=int, default=36)
    ap.add_argument("--lp", type=int, default=8)
    ap.add_argument("--hooks", default="all")
    ap.add_argument("--simd", action="store_true", help="the hooks on the reference's x86 SIMD build (make -f oracle/Makefile.enc simd)")
    ap.add_argument("--numa", action="store_true", help="-ss 0 for the first half of the instances, -ss 1 for the second")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--same-device", action="store_true", help="every instance on device 0: N concurrent encodes sharing ONE GPU")
    ap.add_argument("--reference", action="store_true", help="no hooks: N concurrent instances of the reference encoder (the CPU-only comparison of --same-device)")
    ap.add_argument("--app")
    ap.add_argument("--workdir")
    ap.add_argument("--timeout", type=int, default=1800)
    a = ap.parse_args()
    out = run(a)
    print(json.dumps(out))
    ok = all(i["rc"] == 0 and (a.reference or i["ordinal"] == (0 if a.same_device else i["instance"])) for i in out["instances"]) and all(out.get("identical_to_single_runs", [True]))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
