"""The C port of the oracle as bench.py's cpu_baseline leg runs it: in a child process (oracle/time_port.py), sized to the
CPU quota of the box.  Checked here against the Python oracle on a small input."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_time_port_child_matches_python_oracle(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    from oracle import msm_oracle as O

    C = O.BLS12_377
    n = 1 << 9
    pts, _ = O.random_points_bls377("cpu-port", 64)
    pts = pts * (n // 64)   # repeated points: equal-x pairs inside the buckets
    sc = O.prng_ints("cpu-port/s", n, C.q)
    (tmp_path / "points.bin").write_bytes(O.points_to_bytes(pts, 48))
    (tmp_path / "scalars.bin").write_bytes(O.scalars_to_bytes(sc))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_port.py"), str(tmp_path), "5", "8", "9"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-400:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert [e["log2_n"] for e in rep["series"]] == [8, 9]
    for e in rep["series"]:
        m = 1 << e["log2_n"]
        exp = O.msm_batched_affine(sc[:m], pts[:m], c=7)
        assert (int(e["result"][0], 16), int(e["result"][1], 16)) == exp
        assert e["threads"] >= 1 and len(e["times_s"]) >= 1
    q = c_oracle.load().oracle_cpu_quota()
    assert q == rep["quota"] and 0 <= q <= (os.cpu_count() or 1)
