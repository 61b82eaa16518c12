"""Extracts the 20 x 32 Sobol direction-number table (Joe-Kuo data; public constants, not
code) from the reference's shader/basic/sobol_matrix.glsl and writes it as a C initializer
list used by both the oracle and the device code.  Run once in the build container:
    python tools/gen_sobol_table.py /root/reference
"""
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
text = open(ref + "/shader/basic/sobol_matrix.glsl").read()
body = text.split("#else")[1]
nums = [int(x) for x in re.findall(r"(\d+)u", body.split("{", 1)[1])]
assert len(nums) == 640, len(nums)
# row 0 is the van der Corput row: 2^31, 2^30, ...
assert nums[:32] == [1 << (31 - j) for j in range(32)]
lines = []
for d in range(20):
    lines.append(", ".join("%du" % v for v in nums[d * 32:(d + 1) * 32]) + ",")
out = ("// Sobol direction numbers, 20 dimensions x 32 bits (V[d*32 + j]); data taken from\n"
       "// the reference's shader/basic/sobol_matrix.glsl:25-46 by tools/gen_sobol_table.py.\n"
       + "\n".join(lines) + "\n")
for path in ("oracle/sobol_table.inc", "raytracing_amd/csrc/device/sobol_table.inc"):
    open(path, "w").write(out)
print("ok")
