"""Host-side schedulers (mimo_unet_amd/csrc/tile_sched.h: XCD workgroup order, tile shapes, channel-tile widths, split
counts, the wide-convolution dispatch) compiled with g++ -fsanitize=address,undefined and swept over their argument
ranges — the kernels consume exactly these functions (the .hip files include the same header)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedulers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sched_test")
    src = os.path.join(ROOT, "tests", "host", "sched_test.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-Wall", "-Werror", src, "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "all checks passed" in r.stdout


def test_kernels_use_the_tested_header():
    """No private copy of a scheduler is left in the .hip sources."""
    csrc = os.path.join(ROOT, "mimo_unet_amd", "csrc")
    for fn, needles in (("conv_bf16x3.hip", ["using sched::pick_tile_n"]), ("conv3x3.hip", ["sched::pick_tile_n", "sched::conv_cout_pad"]),
                        ("wgrad_split.hip", ["sched::wg_tiles", "sched::wg_pick_splits"]), ("plan.hip", ["sched::wg_side_cus"]), ("common.h", ["sched::xcd_virtual_index", "sched::w16_scale", "sched::wg_dz_scale"]),
                        ("conv_wide.hip", ["sched::wide_config", "sched::wide_grid_x"])):
        text = open(os.path.join(csrc, fn)).read()
        for n in needles:
            assert n in text, (fn, n)
