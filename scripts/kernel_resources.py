"""VGPR / SGPR / scratch / LDS of every kernel of hip_kernels.hip (device assembly of the gfx950 build):  python scripts/kernel_resources.py [f64|f32] [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
defs = ["-DMAT_VAL_TYPE=double"] if dt == "f64" else ["-DMAT_VAL_TYPE=float", "-DTILESPMV_F32"]
out = "/tmp/kres_%s.s" % dt
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "--offload-arch=gfx950", "-munsafe-fp-atomics", "-w"] + defs + os.environ.get("KRES_DEFS", "").split() +
               ["-S", "--cuda-device-only", os.path.join(root, "tilespmv_amd/csrc/hip_kernels.hip"), "-o", out], check=True)
s = open(out).read()
names = re.findall(r"\.amdhsa_kernel (\S+)", s)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
for (m, d) in zip(re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S), dem):
    body = m.group(2)
    d = d.split("(")[0].replace("void tilespmv::", "")
    if flt and flt not in d:
        continue
    g = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1)
    print("%-70s vgpr %3s sgpr %3s scratch %4s lds %6s" % (d, g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
