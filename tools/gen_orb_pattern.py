#!/usr/bin/env python3
"""Extract the 256x4 rBRIEF sampling table (data, not code) from the reference tree.

The table is the descriptor-compatibility constant of the ORB configuration
(/root/reference/thirdparty/ORBextractor.cpp:150-408).  It is emitted as a bare
comma-separated integer list so that both the oracle and the HIP kernels can
`#include` it inside an array initialiser.  Runs only in the build container
(the reference tree does not exist on the GPU box); the output is committed.
"""
import re, sys
src = open("/root/reference/thirdparty/ORBextractor.cpp").read().splitlines()
body = "\n".join(src[150:408])              # lines 151..408 (1-based), inside the braces
body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
nums = [int(t) for t in re.findall(r"-?\d+", body)]
assert len(nums) == 1024, len(nums)
out = []
for i in range(0, 1024, 16):
    out.append(",".join(str(v) for v in nums[i:i+16]) + ",")
text = "\n".join(out) + "\n"
for path in sys.argv[1:]:
    open(path, "w").write(text)
print("ok", len(nums), "values; first 8:", nums[:8], "last 4:", nums[-4:])
