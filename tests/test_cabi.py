"""CPU: the C-ABI library loads, exports every symbol include/ba_mi355x.h declares, its host-side pieces (BAL loader,
writer, synthetic generator, shard plan) behave like the reference's loader, and the product path refuses to run
without a GPU (no CPU fallback).  No compute entry point is called here."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import DATA21, ROOT, to_oracle

HEADER = os.path.join(ROOT, "include", "ba_mi355x.h")
BIN = os.path.join(ROOT, "bundleadjustment_benchmarks_amd", "bin")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ba_[a-z0-9_]+)\s*\(", src)) - {"ba_trial_cb", "ba_allreduce_fn"})


def test_library_exports_every_declared_symbol(ba):
    L = ba.lib()
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(ba.EXPORTS) <= set(names)


def test_status_and_error_strings(ba):
    # statusToString, BacktrackLevMarqQRChol.h:48-63
    assert ba.status_string(0) == "Success (Energy Flatlined)"
    assert ba.status_string(1) == "Success (Exceeded Maximum Lambda)"
    assert ba.status_string(2) == "Too Many Function Evaluations"
    assert ba.status_string(3) == "Maximum Iterations Reached"
    assert ba.status_string(-1) == "Running" and ba.status_string(-2) == "Not Started"
    assert "open" in ba.error_string(2)


def test_loader_matches_reference_format(ba, O):
    p = ba.Problem.load_bal(DATA21)
    assert (p.N, p.M, p.K) == (21, 11315, 36455)
    a = p.arrays()
    po = O.load_bal(DATA21)  # the oracle's independent fscanf restatement of the reference's `ifs >>` loop
    for k in ("cam_idx", "pt_idx", "meas", "cams9", "pts"):
        assert np.array_equal(a[k], getattr(po, k))
    assert a["cam_idx"][:2].tolist() == [0, 1] and a["meas"][0] == 1.597070e+03  # first data line of the file
    assert np.all(np.diff(a["pt_idx"]) >= 0)


def test_loader_errors(ba, tmp_path):
    with pytest.raises(ba.BAError) as e:
        ba.Problem.load_bal(str(tmp_path / "nope.txt"))
    assert e.value.code == 2  # WrongInputFile
    bad = tmp_path / "bad.txt"
    bad.write_text("2 3 4\n0 0 1.0 2.0\n0 1 oops\n")
    with pytest.raises(ba.BAError) as e:
        ba.Problem.load_bal(str(bad))
    assert e.value.code == 3
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    with pytest.raises(ba.BAError) as e:
        ba.Problem.load_bal(str(empty))
    assert e.value.code == 3
    oob = tmp_path / "oob.txt"
    oob.write_text("1 1 1\n5 0 1.0 2.0\n" + "0.0\n" * 12)
    with pytest.raises(ba.BAError) as e:
        ba.Problem.load_bal(str(oob))
    assert e.value.code == 3
    # an index that would wrap into a valid int (2^32) and non-finite tokens (strtod accepts them) are refused
    for name, text in (("wrap.txt", "1 1 1\n4294967296 0 1.0 2.0\n" + "0.0\n" * 12),
                       ("nan.txt", "1 1 1\n0 0 nan 2.0\n" + "0.0\n" * 12),
                       ("inf.txt", "1 1 1\n0 0 1.0 2.0\n" + "0.0\n" * 11 + "inf\n")):
        f = tmp_path / name
        f.write_text(text)
        with pytest.raises(ba.BAError) as e:
            ba.Problem.load_bal(str(f))
        assert e.value.code == 3, name


def test_save_load_round_trip_and_unsorted_input(ba, tmp_path):
    p = ba.Problem.synthetic(5, 40, 130, 3)
    f = tmp_path / "s.txt"
    p.save_bal(str(f))
    q = ba.Problem.load_bal(str(f))
    a, b = p.arrays(), q.arrays()
    for k in a:
        assert np.array_equal(a[k], b[k])
    # an input that is not sorted by point is accepted (the structure builder sorts it stably)
    perm = np.random.default_rng(0).permutation(p.K)
    u = ba.Problem.from_arrays(p.N, p.M, p.K, a["cam_idx"][perm], a["pt_idx"][perm], a["meas"].reshape(-1, 2)[perm].ravel(),
                               a["cams9"], a["pts"])
    assert u.shard_plan(0, 1)["was_sorted"] == 0 and p.shard_plan(0, 1)["was_sorted"] == 1
    assert u.shard_plan(0, 1)["entries"] == p.shard_plan(0, 1)["entries"]


def test_binary_cache_round_trip(ba, tmp_path):
    p = ba.Problem.load_bal(DATA21)
    f = tmp_path / "p21.bacache"
    p.save_cache(str(f))
    q = ba.Problem.load_cache(str(f))
    a, b = p.arrays(), q.arrays()
    assert (q.N, q.M, q.K) == (p.N, p.M, p.K) and all(np.array_equal(a[k], b[k]) for k in a)
    (tmp_path / "bad.bacache").write_bytes(b"not a cache")
    with pytest.raises(ba.BAError) as e:
        ba.Problem.load_cache(str(tmp_path / "bad.bacache"))
    assert e.value.code == 3
    trunc = tmp_path / "trunc.bacache"
    trunc.write_bytes(f.read_bytes()[:-100])
    with pytest.raises(ba.BAError) as e:
        ba.Problem.load_cache(str(trunc))
    assert e.value.code == 3


def test_synthetic_generator(ba, O):
    p = ba.Problem.synthetic(16, 500, 1800, 42)
    q = ba.Problem.synthetic(16, 500, 1800, 42)
    r = ba.Problem.synthetic(16, 500, 1800, 43)
    a, b, c = p.arrays(), q.arrays(), r.arrays()
    assert all(np.array_equal(a[k], b[k]) for k in a) and not np.array_equal(a["meas"], c["meas"])
    assert (p.N, p.M, p.K) == (16, 500, 1800)
    k = np.bincount(a["pt_idx"], minlength=500)
    assert k.min() >= 2 and k.sum() == 1800
    assert np.all(np.diff(a["pt_idx"]) >= 0)
    for j in (0, 17, 499):  # cameras of one point are distinct and sorted
        cams = a["cam_idx"][a["pt_idx"] == j]
        assert np.all(np.diff(cams) > 0)
    po = to_oracle(p)
    st = O.stats(po, O.init_cams(po), po.pts)
    assert 0.3 < st["mean_err"] < 10 and st["n_inliers"] > 0.05 * p.K  # a solvable, BAL-like start
    with pytest.raises(ba.BAError):
        ba.Problem.synthetic(4, 100, 150, 1)  # K < 2M


def test_shard_plan_partitions_points(ba):
    p = ba.Problem.load_bal(DATA21)
    full = p.shard_plan(0, 1)
    assert (full["p0"], full["p1"], full["o0"], full["o1"]) == (0, p.M, 0, p.K)
    assert full["pairs"] == 21 * 22 // 2
    for world in (2, 3, 8):
        plans = [p.shard_plan(r, world) for r in range(world)]
        assert plans[0]["p0"] == 0 and plans[-1]["p1"] == p.M and plans[-1]["o1"] == p.K
        for a, b in zip(plans, plans[1:]):
            assert a["p1"] == b["p0"] and a["o1"] == b["o0"]
        counts = [q["o1"] - q["o0"] for q in plans]
        assert max(counts) - min(counts) <= 2 * 15 + 1  # balanced by observation count (max 15 obs per point)
        assert sum(q["entries"] for q in plans) == full["entries"]


def test_no_gpu_no_fallback(ba):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = ba.Problem.synthetic(4, 30, 100, 1)
    with pytest.raises(ba.BAError) as e:
        ba.Solver(p)
    assert e.value.code == 5  # BA_ERR_HIP: the product path never falls back to a CPU


@pytest.mark.parametrize("exe", ["Bundle_Adjustment_QRKit", "Bundle_Adjustment_QRChol", "Bundle_Adjustment_Cholesky",
                                 "Bundle_Adjustment_Cholesky_f32", "Bundle_Adjustment_MoreQR", "Bundle_Adjustment_MoreQR_f32",
                                 "Bundle_Adjustment_SPQR", "Bundle_Adjustment_QRKit_f32"])
def test_executables_keep_reference_cli(exe):
    """Usage / exit codes of the reference driver (bundle_adjustment_large.cpp:26-31,45-54)."""
    path = os.path.join(BIN, exe)
    assert os.path.exists(path), "run __graft_entry__.build()"
    r = subprocess.run([path], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr and "<sparse reconstruction file>" in r.stderr
    r = subprocess.run([path, "/nonexistent/file.txt"], capture_output=True, text=True)
    assert r.returncode == 2 and r.stderr.strip() == "Cannot open /nonexistent/file.txt"


def test_shard_plan_with_more_ranks_than_points(ba):
    p = ba.Problem.synthetic(3, 5, 12, 2)
    plans = [p.shard_plan(r, 8) for r in range(8)]
    assert plans[0]["p0"] == 0 and plans[-1]["p1"] == 5
    assert sum(q["o1"] - q["o0"] for q in plans) == 12 and sum(q["p1"] - q["p0"] for q in plans) == 5
    assert any(q["p1"] == q["p0"] for q in plans)  # some shards are empty: allowed


# ---- rendezvous file of a sharded launch (ba_comm_id_via_file, reader side) ---------------------------------------------
# The file format is the documented one (128-byte id + 8-byte nonce hash, 0 without BA_COMM_NONCE), so a test can play rank 0 without
# RCCL.  Each case runs in a fresh process: "process start" is the moment the library is loaded.

_READER = r"""
import ctypes, os, sys, time
sys.path.insert(0, %(root)r)
import bundleadjustment_benchmarks_amd as ba
L = ba.lib()                      # <- the reader's process start, as the library sees it
time.sleep(float(sys.argv[2]))    # problem load + solver creation of a big shard
buf = ctypes.create_string_buffer(128)
t0 = time.time()
rc = L.ba_comm_id_via_file(sys.argv[1].encode(), 1, buf)
print(rc, "%%.2f" %% (time.time() - t0), buf.raw[:4].hex())
"""


def _reader(tmp_path, path, delay, env=None):
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "reader.py"
    script.write_text(_READER % {"root": ROOT})
    e = dict(os.environ, BA_COMM_WAIT_S="12")
    e.pop("BA_COMM_NONCE", None)
    e.update(env or {})
    return subprocess.Popen([sys.executable, str(script), str(path), str(delay)], stdout=subprocess.PIPE, text=True, env=e)


def _publish(path, first4, nonce=0):
    tmp = str(path) + ".t"
    with open(tmp, "wb") as f:
        f.write(bytes(first4) + bytes(124) + int(nonce).to_bytes(8, "little"))
    os.replace(tmp, path)


def test_comm_file_written_before_a_slow_readers_call_is_accepted_at_once(tmp_path):
    """Rank 0 publishes; the reader reaches ba_comm_id_via_file 3 s later (its shard took that long to load).  Round 3 measured the
    file's age against the CALL and refused this launch's valid file for 60 s; the yardstick is the reader's process start."""
    import time
    path = tmp_path / "comm.id"
    p = _reader(tmp_path, path, 3.0)
    time.sleep(1.0)  # (the reader process is up and has loaded the library by now: torch is not imported, ~0.3 s)
    _publish(path, b"\x01\x02\x03\x04")
    rc, waited, head = p.communicate(timeout=60)[0].split()
    assert int(rc) == 0 and head == "01020304" and float(waited) < 1.0, (rc, waited, head)


def test_comm_file_left_by_a_dead_run_is_not_taken_while_rank0_replaces_it(tmp_path):
    import time
    path = tmp_path / "comm.id"
    _publish(path, b"\xde\xad\xde\xad")
    os.utime(path, (time.time() - 100, time.time() - 100))
    p = _reader(tmp_path, path, 0.0)
    time.sleep(2.0)  # inside the 5 s grace period: rank 0 of this launch starts, removes the leftover, publishes its own id
    os.unlink(path)
    time.sleep(0.3)
    _publish(path, b"\x0a\x0b\x0c\x0d")
    rc, waited, head = p.communicate(timeout=60)[0].split()
    assert int(rc) == 0 and head == "0a0b0c0d", (rc, waited, head)


def test_comm_file_of_an_early_rank0_is_taken_after_the_grace_period(tmp_path):
    """Rank 0 started by hand long before this reader: the file is older than the reader's process, nobody replaces it -- accepted once
    it has stayed unchanged for BA_COMM_GRACE_S."""
    import time
    path = tmp_path / "comm.id"
    _publish(path, b"\x11\x22\x33\x44")
    os.utime(path, (time.time() - 30, time.time() - 30))
    p = _reader(tmp_path, path, 0.0, env={"BA_COMM_GRACE_S": "2"})
    rc, waited, head = p.communicate(timeout=60)[0].split()
    assert int(rc) == 0 and head == "11223344" and 1.5 < float(waited) < 6.0, (rc, waited, head)


def test_comm_file_with_another_launchs_nonce_is_never_taken(tmp_path):
    path = tmp_path / "comm.id"
    _publish(path, b"\x55\x55\x55\x55", nonce=12345)
    p = _reader(tmp_path, path, 0.0, env={"BA_COMM_NONCE": "this-launch", "BA_COMM_WAIT_S": "3"})
    rc, waited, head = p.communicate(timeout=60)[0].split()
    assert int(rc) != 0 and float(waited) >= 2.9
