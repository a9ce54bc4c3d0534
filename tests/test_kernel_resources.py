"""What fits beside k_march is part of the design: the marching kernel holds
four waves on every SIMD and most of every CU's LDS while panoramas overlap, and a kernel of
another stream that does not fit into what is left waits for a marching wave to retire - k_clip
once waited 0.7 ms of every panorama that way, and a k_march with 8 registers more cost 7 %.
The footprints are read from the code object inside the built library (no GPU needed)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels():
    lib = os.path.join(ROOT, "horizonator_amd", "libhorizonator.so")
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(lib) and os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("library or llvm tools not present")
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)                      # (the tool writes the bundles next to its input)
        subprocess.run([objdump, "--offloading", copy], check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert len(co) == 1, os.listdir(tmp)
        notes = subprocess.run([readelf, "--notes", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count")[1:]:
        f = {k: v for k, v in re.findall(r"\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)", block)}
        out[f["name"]] = {k: int(v) for k, v in f.items() if k != "name"}
    return out


def _one(kernels, fragment):
    hit = [v for k, v in kernels.items() if fragment in k]
    assert len(hit) == 1, (fragment, [k for k in kernels if fragment in k])
    return hit[0]


def test_what_runs_beside_the_marching_kernel_fits_beside_it():
    k = _kernels()
    march = _one(k, "k_marchILb0ELb0ELb0ELb0E")     # the production instance (no per-wave counters, no coarse depth, every vertex computed, one queue counter)
    assert march["private_segment_fixed_size"] == 0, "k_march<false, false, false, false> must not use scratch memory"
    zoomed = _one(k, "k_marchILb0ELb1ELb0ELb0E")    # ... of series (coarse depth, hz_k_hiz.h)
    # ... those that read the vertex cache (round 5), and those of zoomed views (sixteen queue counters: hz_types.h, HZ_QSHARDS)
    for name in ("k_marchILb0ELb0ELb1ELb0E", "k_marchILb0ELb1ELb1ELb0E", "k_marchILb0ELb0ELb0ELb1E", "k_marchILb0ELb1ELb0ELb1E", "k_marchILb0ELb0ELb1ELb1E", "k_marchILb0ELb1ELb1ELb1E"):
        cached = _one(k, name)
        assert cached["private_segment_fixed_size"] == 0 and cached["vgpr_count"] <= 104 and cached["group_segment_fixed_size"] <= 7168, (name, cached)
    assert zoomed["private_segment_fixed_size"] == 0 and zoomed["vgpr_count"] <= 112 and zoomed["group_segment_fixed_size"] <= 7168, zoomed
    # round 4: 104 registers at most - four marching waves then leave 96 of a SIMD's 512, i.e. TWO waves of k_big or of the
    # conversion (48 each) instead of one: the first round's k_big beside a marching kernel 650 -> 450 us, a render of a
    # series 0.93 -> 0.85 ms (profiles/r4_ab_kbig_addr32.txt)
    assert march["vgpr_count"] <= 104 and zoomed["vgpr_count"] <= 104, (march, zoomed)
    # round 5: ... and more than 96 - FOUR waves per SIMD, not five, which would leave their neighbours no room at all (a series
    # of renders 0.85 -> 0.98 ms with a 96-register build of the kernel: profiles/r5_ab_march_loop.txt)
    for name in [f"k_marchILb0ELb{h}ELb{v}ELb{q}E" for h in (0, 1) for v in (0, 1) for q in (0, 1)]:
        assert 96 < _one(k, name)["vgpr_count"] <= 104, (name, _one(k, name))
    assert march["group_segment_fixed_size"] <= 7168
    left_vgprs = 512 - 4*((march["vgpr_count"] + 7)//8*8)
    left_lds = 160*1024 - 16*march["group_segment_fixed_size"]
    # (k_pack_host: a call that delivers into host memory converts sector s while the marching kernel of sector s+1 runs - round 5)
    for name in ("k_bigILb0E", "k_bigILb1E", "k_resolve4ILb1E", "k_resolve4ILb0E", "k_pack_hostILb1E", "k_pack_hostILb0E", "k_clipILb0E", "k_clipILb1E"):
        other = _one(k, name)
        assert other["private_segment_fixed_size"] == 0, name
        # two waves of each beside the marching waves (k_clip: one, when a marching wave has gone)
        assert 2*((other["vgpr_count"] + 7)//8*8) <= left_vgprs or name.startswith("k_clip"), (name, other, left_vgprs)
        assert other["group_segment_fixed_size"] <= 5120 and 2*other["group_segment_fixed_size"] <= left_lds, (name, other)
    # k_clip is a 64-thread workgroup that is placed when one marching wave has gone: 96 registers at most
    assert _one(k, "k_clipILb0E")["vgpr_count"] <= 96 and _one(k, "k_clipILb1E")["vgpr_count"] <= 96
