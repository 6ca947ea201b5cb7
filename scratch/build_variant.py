"""Build scratch/variants/<name>.so = the product library with ONE source recompiled under extra -D flags:
   python scratch/build_variant.py <name> <source.hip> -DX=1 -DY=0 ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import build as B
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_library()
os.makedirs(os.path.join(os.path.dirname(__file__), "variants"), exist_ok=True)
obj = os.path.join(B.OBJ, "_var_%s.o" % name)
cmd = ["hipcc"] + B.COMMON + B.EXTRA.get(src, []) + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj]
subprocess.run(cmd, check=True)
objs = [obj if s == src else os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B._sources()]
out = os.path.join(os.path.dirname(__file__), "variants", name + ".so")
subprocess.run(["hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs, check=True)
print("built", out)
