"""GPU test (-m gpu): the C ABI driven from PLAIN C (tests/cabi_driver.c, gcc, linked against libpioran_hip.so) on the
reference's literal N = 6 series (test/test_scalablegp.jl:109-132) — no Python marshalling between the caller and the
library.  The driver runs as a child process; expected values come from the oracle."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
DRIVER = ROOT / "tests" / "cabi_driver"


def build_driver():
    src = ROOT / "tests" / "cabi_driver.c"
    lib = ROOT / "pioran.jl_amd" / "libpioran_hip.so"
    if DRIVER.exists() and DRIVER.stat().st_mtime >= max(src.stat().st_mtime, lib.stat().st_mtime):
        return
    subprocess.run(["gcc", "-std=c11", "-O1", f"-I{ROOT / 'include'}", str(src), "-o", str(DRIVER), f"-L{ROOT / 'pioran.jl_amd'}",
                    "-lpioran_hip", "-Wl,-rpath,$ORIGIN/../pioran.jl_amd", "-lm"], check=True)


def test_plain_c_driver(tmp_path, golden_dir):
    import pioran_jl_amd as pj
    from oracle import oracle as O
    build_driver()
    g = json.loads((golden_dir / "reference_literals.json").read_text())["scalablegp_n6"]
    t, y, yerr = np.array(g["t"]), np.array(g["y"]), np.array(g["yerr"])
    rows = []
    for i in range(10):
        a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, g["alpha1"][i], g["f1"][i], g["alpha2"][i]),
                              g["f_min"], g["f_max"], g["n_components"], g["variance"][i])
        mu, nu = g["mu"][i], 1.0 + 0.05 * i
        rows.append((a, b, mu, nu, O.logl(a, b, c, d, t, y - mu, nu * yerr ** 2)))
    dense = O.dense_nll(rows[0][0], rows[0][1], c, d, t, y - rows[0][2], rows[0][3] * yerr ** 2)
    J = len(c)
    fmt = lambda v: " ".join(repr(float(x)) for x in np.atleast_1d(v))
    lines = [f"{J} {len(t)} {len(rows)}", fmt(c), fmt(d), fmt(t), fmt(y), fmt(yerr ** 2)]
    lines += [f"{fmt(a)} {fmt(b)} {mu!r} {nu!r} {val!r}" for a, b, mu, nu, val in rows]
    lines.append(repr(float(dense)))
    case = tmp_path / "case.txt"
    case.write_text("\n".join(lines) + "\n")
    r = subprocess.run([str(DRIVER), str(case)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CABI OK" in r.stdout, r.stdout + r.stderr
