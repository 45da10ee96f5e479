"""What the gfx950 compiler made of hip_kernels.hip, checked without a GPU: no kernel of either build touches scratch memory (a spilled register costs a kernel more than
any instruction it saves), and the number of kernel instantiations stays bounded (round 6 retired the x-window and slab-pacing template axes and made the XCD remap a
run-time branch: 144 -> 91 per value type).  Reads the device assembly (`hipcc -S --cuda-device-only`), the same source and flags as tilespmv_amd/csrc/Makefile."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_KERNELS = 96


def _device_asm(dt, out):
    defs = ["-DMAT_VAL_TYPE=double"] if dt == "f64" else ["-DMAT_VAL_TYPE=float", "-DTILESPMV_F32"]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "--offload-arch=gfx950", "-munsafe-fp-atomics", "-w"] + defs +
                   ["-S", "--cuda-device-only", os.path.join(ROOT, "tilespmv_amd/csrc/hip_kernels.hip"), "-o", out], check=True)
    return open(out).read()


def test_no_kernel_spills_and_the_instantiation_count_is_bounded(tmp_path):
    with ThreadPoolExecutor(2) as ex:
        asm = dict(zip(("f64", "f32"), ex.map(lambda dt: _device_asm(dt, str(tmp_path / (dt + ".s"))), ("f64", "f32"))))
    for dt, s in asm.items():
        kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S)
        assert 40 <= len(kernels) <= MAX_KERNELS, (dt, len(kernels))
        spills = {name: int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) for name, body in kernels}
        assert not {k: v for k, v in spills.items() if v}, (dt, {k: v for k, v in spills.items() if v})
        assert ("v_mfma_f64_16x16x4_f64" if dt == "f64" else "v_mfma_f32_16x16x4_f32") in s      # dense tiles run on the matrix cores in both builds
