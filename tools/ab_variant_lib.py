"""Same-box A/B of a compile-time switch: builds a second library (tools/_ab/libpioran_hip_<tag>.so) that differs from the shipped one by -D flags on ONE
source, links it with the shipped objects of the others.  `python tools/ab_variant_lib.py <tag> <source.hip> -DNAME=VALUE ...`, then on the GPU box
`PIORAN_HIP_LIB=tools/_ab/libpioran_hip_<tag>.so python bench.py ...` next to the plain command (tools/ab_variant_run.sh)."""
import subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import importlib.util
spec = importlib.util.spec_from_file_location("pbuild", Path(__file__).resolve().parents[1] / "pioran.jl_amd" / "build.py")
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
tag, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build(verbose=False)
out = Path(__file__).resolve().parent / "_ab"; out.mkdir(exist_ok=True)
obj = out / f"{Path(src).stem}_{tag}.o"
cc = b.hipcc()
subprocess.run([cc, *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *defs, "-c", str(b.CSRC / src), "-o", str(obj)], check=True)
objs = [str(obj) if s == src else str(b.OBJ / (Path(s).stem + ".o")) for s in b.SOURCES]
lib = out / f"libpioran_hip_{tag}.so"
subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *objs], check=True)
print(lib)
