"""The C++ host side above the C-ABI: hip_accel.hpp (models the reference's `accelerator` concept) and the
rtk_render CLI (the reference's src/main.cpp with runtime flags)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENE5

PKG = os.path.join(ROOT, "simd-raytracer_amd")
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")


def _build_adapter_check():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "adapter_check")
    src = os.path.join(ROOT, "tests", "cpp", "adapter_check.cpp")
    deps = [src, os.path.join(PKG, "hip_accel.hpp"), os.path.join(ROOT, "include", "rtk.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call([
            "g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "cpp", "mock"),
            "-I" + os.path.join(ROOT, "include"), "-I" + PKG, src, "-o", exe, "-L" + PKG, "-lrtk_hip",
            "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return exe


def test_adapter_compiles_and_models_the_accelerator_concept(rtk):
    """static_assert(accelerator<hip_accel<float, eps>, float>) inside adapter_check.cpp is the actual check."""
    assert os.path.exists(_build_adapter_check())


@pytest.mark.gpu
def test_adapter_results(rtk, ora):
    out = subprocess.run([_build_adapter_check()], capture_output=True, text=True, check=True).stdout
    single = re.search(r"single hit=1 t=(\S+) u=(\S+) v=(\S+) w=(\S+) mesh=0 n=\((\S+),(\S+),(\S+)\) pos=\((\S+),(\S+),(\S+)\) miss=1", out)
    assert single, out
    t, u, v, w, nx, ny, nz, px, py, pz = map(float, single.groups())
    assert t == 3.0 and (px, py, pz) == (0.0, 0.0, -3.0)
    assert abs(u + v + w - 1.0) < 1e-6 and (nx, ny, nz) == (0.0, 0.0, 1.0)
    assert "batch 1 0 1" in out
    assert re.search(r"frame 16x16 rays=\d+ centre=\(\S+\) corner=\(0,0.5,0\)", out), out


@pytest.mark.gpu
def test_rtk_render_cli_writes_the_reference_ppm(rtk, ora, tmp_path):
    exe = os.path.join(PKG, "rtk_render")
    out = tmp_path / "image.ppm"
    res = subprocess.run([exe, SCENE5, "--width", "320", "--height", "180", "--out", str(out)], capture_output=True,
                         text=True, check=True)
    assert re.search(r"Rendering took \S+ seconds\.", res.stdout)
    ref, cn = ora.Accel(ora.Scene(ora.load_crtscene(SCENE5)), ora.ACCEL_KD_SIMD).render(320, 180, 1, 5, 0)
    assert out.read_bytes() == ora.write_ppm(ref)
    assert f"{cn['rays']} rays" in res.stdout


@pytest.mark.gpu
def test_rtk_render_cli_world_path_through_rccl(rtk, ora, tmp_path):
    """`rtk_render --world N`: the launcher forks one rank process per GPU before touching the device; a rank renders its
    buckets, the bucket buffers go through an RCCL all-gather, the frame is assembled on the device.  A one-GPU box can only
    run world 1 (RCCL refuses two ranks on one device), which still goes through the fork, the id exchange, communicator
    setup, the collectives and the device-resident frame."""
    exe = os.path.join(PKG, "rtk_render")
    out = tmp_path / "world.ppm"
    res = subprocess.run([exe, SCENE5, "--width", "320", "--height", "180", "--world", "1", "--frames", "3", "--fov", "75", "--out", str(out)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    ref, cn = ora.Accel(ora.Scene(ora.load_crtscene(SCENE5)), ora.ACCEL_KD_SIMD).render(320, 180, 1, 5, 0, fov_degrees=75.0)
    assert out.read_bytes() == ora.write_ppm(ref)
    assert f"{cn['rays']} rays on 1 GPUs" in res.stdout


@pytest.mark.gpu
def test_rtk_render_cli_refuses_more_ranks_than_devices_and_does_not_hang(rtk, tmp_path):
    """`--world N` with N > devices: every rank sees it and leaves; the launcher reports failure instead of waiting for ranks that
    would block in RCCL.  The RCCL id travels through pipes, so nothing is left under /tmp either."""
    import torch

    n = torch.cuda.device_count()
    exe = os.path.join(PKG, "rtk_render")
    before = set(os.listdir("/tmp"))
    res = subprocess.run([exe, SCENE5, "--width", "64", "--height", "36", "--world", str(n + 1), "--out", str(tmp_path / "x.ppm")],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and f"--world {n + 1} but only {n} HIP device" in res.stderr
    assert not [f for f in set(os.listdir("/tmp")) - before if f.startswith("rtk_nccl_id")]
