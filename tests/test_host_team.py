"""The host worker team ppo_ba_create / ppo_ba_read fork their counting passes on (csrc/ppo_pool.hpp, HostTeam), on the CPU:
tests/host_team_test.cpp built with g++, once plain and once under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "point-plane-object-slam_amd", "csrc")


@pytest.mark.parametrize("sanitize", [False, True])
def test_host_team_runs_every_task_once_and_is_race_free(tmp_path, sanitize):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / "host_team_test")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + CSRC,
           os.path.join(HERE, "host_team_test.cpp"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    if sanitize:
        cmd.insert(1, "-fsanitize=thread")
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
    assert "ThreadSanitizer" not in out.stderr, out.stderr


def test_device_block_cache_slab_and_staging_ownership(tmp_path):
    """tests/pool_test.cpp: the slab ppo_ba_warmup reserves for the device block cache (pointer-bump misses, blocks in the ordinary free lists, the slab
    released only when nobody holds a block of it) and Stage::owns_pinned (which uploads may be batched into the copy kernel), on the CPU."""
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / "pool_test")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + CSRC,
           os.path.join(HERE, "pool_test.cpp"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "pool_test OK" in out.stdout, out.stdout + out.stderr
