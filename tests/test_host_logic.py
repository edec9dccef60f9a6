"""CPU tests of the host side: C-ABI surface, dictionary handling, axis swapping, geometry bookkeeping.
(No compute entry point is called here: that needs a GPU and lives in the -m gpu tests.)"""
import ctypes as C
import os
import re

import numpy as np
import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from tomobar_amd import _lib
    header = open(os.path.join(ROOT, "include", "tomo_mi355x.h")).read()
    declared = set(re.findall(r"\b(tomo_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    handle = _lib.lib()
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/tomo_mi355x.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    # the header's number, the binding's number and the built library agree (a stale .so fails at load, not later)
    hdr_ver = int(re.search(r"#define\s+TOMO_ABI_VERSION\s+(\d+)", header).group(1))
    assert handle.tomo_abi_version() == hdr_ver == _lib.ABI_VERSION
    assert C.sizeof(_lib.AngleRecord) == 32


def test_shipped_library_carries_no_ab_variants_or_probes():
    """the shipped library accepts only its defaults (+ the opt-in exact PD_TV roundings, 22); the A/B
    variants and the measurement switches exist in libtomo_mi355x_dev.so alone.  (No kernel runs: tomo_set_variant is host state.)"""
    from tomobar_amd import _lib
    assert _lib.flavour() == "shipped"
    ship = _lib.lib()
    assert ship.tomo_build_flavour() == b"shipped"
    for name, good, bad in (("bp", (0,), (1, 2, 3, 7)), ("fp", (0,), (1, 2, 3)), ("pdtv", (0, 22), (1, 2, 3, 21, 31, 33)),
                            ("roftv", (0,), (1, 2, 3, 4))):
        for v in bad:
            assert ship.tomo_set_variant(name.encode(), v) == _lib.E_INVALID, (name, v)
            assert b"libtomo_mi355x_dev.so" in ship.tomo_last_error()
        for v in good:
            assert ship.tomo_set_variant(name.encode(), v) == _lib.OK, (name, v)
        assert ship.tomo_set_variant(name.encode(), 0) == _lib.OK
    assert ship.tomo_set_variant(b"probe", 1) == _lib.E_INVALID
    with _lib.use_flavour("dev") as dev:
        assert _lib.flavour() == "dev" and dev is _lib.lib() and dev is not ship
        assert dev.tomo_build_flavour() == b"dev"
        for name, vs in (("bp", (1, 2, 3)), ("fp", (1, 2, 3)), ("pdtv", (1, 2, 3, 21, 22, 31, 32)), ("roftv", (1, 2, 3, 4)), ("probe", (1, 0))):
            for v in vs:
                assert dev.tomo_set_variant(name.encode(), v) == _lib.OK, (name, v)
            dev.tomo_set_variant(name.encode(), 0)
        assert dev.tomo_set_variant(b"pdtv", 33) == _lib.E_INVALID
    assert _lib.flavour() == "shipped" and _lib.lib() is ship
    # the shipped library really is the smaller build
    assert os.path.getsize(_lib.LIB_PATHS["shipped"]) < os.path.getsize(_lib.LIB_PATHS["dev"])


def test_placement_controls_are_host_state():
    """Arena placement search (include/tomo_mi355x.h, tomo_set_placement_tries / tomo_placement_tries / tomo_placement_last):
    the try count is validated host state (the test puts back what it found), and as long as no TV arena of >= 1 GiB was
    placed in this process the report is empty.  Independent of test order: after the GPU tests a report exists and is
    only checked for consistency.  (No kernel runs here.)"""
    import ctypes as C
    from tomobar_amd import _lib, ops
    L = _lib.lib()
    before = L.tomo_placement_tries()
    assert 1 <= before <= 10
    try:
        for bad in (0, -1, 11, 100):
            assert L.tomo_set_placement_tries(bad) == _lib.E_INVALID, bad
            assert b"placement tries" in L.tomo_last_error()
            assert L.tomo_placement_tries() == before
        for good in (1, 10, 6):
            assert L.tomo_set_placement_tries(good) == _lib.OK
            assert L.tomo_placement_tries() == good == ops.placement_tries()
    finally:
        assert L.tomo_set_placement_tries(before) == _lib.OK
    nbytes, chosen, scores = C.c_size_t(7), C.c_int(7), (C.c_double * 16)()
    n = L.tomo_placement_last(C.byref(nbytes), C.byref(chosen), scores, 16)
    if n == 0:
        assert nbytes.value == 0 and chosen.value == -1 and L.tomo_placement_last_fast() == -1
        assert ops.placement_last() is None
    else:
        assert 0 <= chosen.value < n <= 10 and nbytes.value >= 1 << 30 and L.tomo_placement_last_fast() in (0, 1)
        assert ops.placement_last()["scores_GBps"][chosen.value] == max(ops.placement_last()["scores_GBps"])
    assert L.tomo_placement_last(None, None, None, 0) == n


def test_use_flavour_is_per_thread():
    """`with use_flavour("dev")` in one thread must not redirect another thread's ops.* calls."""
    import threading
    from tomobar_amd import _lib
    seen, gate, done = {}, threading.Event(), threading.Event()

    def other():
        gate.wait(10)
        seen["flavour"], seen["lib"] = _lib.flavour(), _lib.lib().tomo_build_flavour()
        done.set()

    t = threading.Thread(target=other)
    t.start()
    ctx = _lib.use_flavour("dev")          # constructing the object changes nothing yet
    assert _lib.flavour() == "shipped"
    with ctx:
        assert _lib.flavour() == "dev"
        gate.set()
        assert done.wait(10)
    t.join()
    assert seen == {"flavour": "shipped", "lib": b"shipped"}
    assert _lib.flavour() == "shipped"
    with pytest.raises(ValueError):
        _lib.use_flavour("nightly")


def test_committed_pmc_traffic_is_keyed_to_the_kernel_sources_at_head():
    """profiles/pmc_traffic.json (what bench.py reports as roofline.traffic_committed) is only valid for the kernel sources the
    counters were taken on: every entry carries the sha of those sources, and this test fails as soon as one of them differs
    from the sources in the tree -- re-run `bash tools/run_pmc_refresh.sh <tag>` (or tools/run_full_set.sh) on the GPU and
    commit the refreshed file together with the kernel change.  (Round 4's figure went stale silently.)"""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    stale = {}
    for key, kern in (("pdtv", "pdtv"), ("pdtv_half", "pdtv"), ("roftv", "roftv"), ("bp", "bp"), ("fp", "fp")):
        assert key in pmc, f"profiles/pmc_traffic.json has no entry for {key}"
        ent = pmc[key]
        assert ent["traffic_bytes"] > 0 and ent.get("profile", "").startswith("profiles/")
        if ent["sources_sha16"] != bench.source_hash(kern):
            stale[key] = (ent["sources_sha16"], bench.source_hash(kern))
    assert not stale, f"stale PMC traffic entries (measured-on sha, current sha): {stale}"
    # the evidence file an entry names must exist in the tree
    for key in ("pdtv", "bp"):
        assert os.path.exists(os.path.join(ROOT, pmc[key]["profile"])), pmc[key]["profile"]


def test_no_gpu_means_loud_failure():
    """The product path has no CPU fallback: without a device every constructor raises."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tomobar_amd import _lib
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    with pytest.raises(_lib.TomoRuntimeError):
        RecToolsIRCuPy(16, 0, 4, 0.0, np.linspace(0, np.pi, 8), 16)
    n = C.c_int(-1)
    assert _lib.lib().tomo_device_count(C.byref(n)) == _lib.E_NODEVICE
    assert b"no CPU fallback" in _lib.lib().tomo_last_error()


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "tomobar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            text = open(os.path.join(dirpath, f), errors="ignore").read() if f.endswith((".py", ".hip", ".h", ".inl")) else ""
            if f.endswith(".py"):
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "libtomo_oracle" not in text, f
            else:  # native sources may cite the oracle in comments but must not include or call it
                assert not re.search(r"#include[^\n]*oracle", text) and not re.search(r"\borc_[a-z0-9_]+\s*\(", text), f


def test_swap_data_axes_tuples():
    # expected tuples are the reference's own (tests/test_tools.py:36-68)
    from tomobar_amd.supp.funcs import _swap_data_axes_to_accepted
    req = ["angles", "detY", "detX"]
    assert _swap_data_axes_to_accepted(["angles", "detX", "detY"], req) == [(1, 2), None]
    assert _swap_data_axes_to_accepted(["detX", "angles", "detY"], req) == [(0, 1), (1, 2)]
    assert _swap_data_axes_to_accepted(["detY", "angles", "detX"], req) == [(0, 1), None]
    assert _swap_data_axes_to_accepted(["angles", "detY", "detX"], req) == [None, None]
    with pytest.raises(ValueError):
        _swap_data_axes_to_accepted(["angles", "detZ", "detX"], req)
    with pytest.raises(ValueError):
        _swap_data_axes_to_accepted(["angles", "detX"], req)


class _FakeTools:
    device_index = 0


class _FakeSelf:
    def __init__(self, os_number=1):
        self.OS_number = os_number
        self.Atools = _FakeTools()


@pytest.fixture
def cpu_ops(monkeypatch):
    """dicts_check moves the data to the GPU through ops.to_device; keep it on the CPU for these host-logic tests."""
    from tomobar_amd import ops
    monkeypatch.setattr(ops, "to_device", lambda x, d=0: torch.as_tensor(x))
    monkeypatch.setattr(ops, "contiguous", lambda t: t.contiguous())
    return ops


def test_dicts_check_defaults_match_reference(cpu_ops):
    # tomobar/supp/dicts.py:83-183 of the reference
    from tomobar_amd.supp.dicts import dicts_check
    data = np.zeros((4, 6, 5), np.float32)
    me = _FakeSelf()
    d, a, r = dicts_check(me, {"projection_data": data}, None, None, "FISTA")
    assert d["data_fidelity"] == "LS" and d["data_axes_labels_order"] is None and me.data_fidelity == "LS"
    assert a == {"iterations": 400, "initialise": None, "nonnegativity": False, "recon_mask_radius": 1.0,
                 "tolerance": 0.0, "verbose": False}
    assert r == {"method": None, "regul_param": 0.001, "iterations": 150, "tolerance": 0.0, "time_marching_step": 0.005,
                 "PD_LipschitzConstant": 12.0, "methodTV": 0, "device_regulariser": 0}
    assert me.nonneg_regul == 0
    _, a, _ = dicts_check(_FakeSelf(5), {"projection_data": data}, None, None, "FISTA")
    assert a["iterations"] == 20
    _, a, _ = dicts_check(_FakeSelf(), {"projection_data": data}, {"nonnegativity": True}, None, "ADMM")
    assert (a["iterations"], a["ADMM_rho_const"], a["ADMM_relax_par"]) == (400, 1.0, 1.6)
    _, a, _ = dicts_check(_FakeSelf(3), {"projection_data": data}, None, None, "ADMM")
    assert a["iterations"] == 10
    for method, its in (("SIRT", 200), ("CGLS", 30), ("Landweber", 1500)):
        _, a, r = dicts_check(_FakeSelf(), {"projection_data": data}, None, None, method)
        assert a["iterations"] == its and a["lipschitz_const"] == 0 and a["tau_step_lanweber"] == 1e-05
        assert r == {"method": None}
    _, a, _ = dicts_check(_FakeSelf(4), {"projection_data": data}, None, None, "OSEM")
    assert a["iterations"] == 15
    # the caller's dictionaries are populated in place, unknown keys are kept
    mine = {"projection_data": data, "mask_diameter": 0.9}
    dicts_check(_FakeSelf(), mine, None, None, "FISTA")
    assert mine["data_fidelity"] == "LS" and mine["mask_diameter"] == 0.9


def test_dicts_check_errors_and_axis_swap(cpu_ops):
    from tomobar_amd.supp.dicts import dicts_check
    data = np.arange(4 * 6 * 5, dtype=np.float32).reshape(4, 6, 5)
    with pytest.raises(NameError):
        dicts_check(_FakeSelf(), None)
    with pytest.raises(NameError):
        dicts_check(_FakeSelf(), {"projection_data": None})
    with pytest.raises(ValueError):
        dicts_check(_FakeSelf(), {"projection_data": data, "data_fidelity": "Huber"})
    with pytest.raises(ValueError):
        dicts_check(_FakeSelf(), {"projection_data": data}, {"nonnegativity": 3})
    for method in ("SIRT", "CGLS", "Landweber"):
        with pytest.raises(NameError):
            dicts_check(_FakeSelf(2), {"projection_data": data}, None, None, method)
    # [angles, detY, detX] -> canonical [detY, angles, detX], materialised contiguous
    d, _, _ = dicts_check(_FakeSelf(), {"projection_data": data, "data_axes_labels_order": ["angles", "detY", "detX"]})
    assert tuple(d["projection_data"].shape) == (6, 4, 5) and d["projection_data"].is_contiguous()
    assert np.array_equal(d["projection_data"].numpy(), np.swapaxes(data, 0, 1))
    # 2D input becomes one detector row
    d, _, _ = dicts_check(_FakeSelf(), {"projection_data": data[0], "data_axes_labels_order": ["detX", "angles"]})
    assert tuple(d["projection_data"].shape) == (1, 5, 6)


def test_check_if_input_2d_or_3d_cpu():
    # tests/test_regularisers.py:7-36 of the reference
    from tomobar_amd.regularisersCuPy import _check_if_input_2d_or_3d
    for shape, want in (((100, 100), ((100, 100), True, 0)), ((10, 100, 100), ((10, 100, 100), False, 0)),
                        ((1, 100, 100), ((100, 100), True, 0)), ((16, 1, 100), ((16, 100), True, 1))):
        d, flag, ax = _check_if_input_2d_or_3d(torch.zeros(shape))
        assert (tuple(d.shape), flag, ax) == want
    with pytest.raises(ValueError):
        _check_if_input_2d_or_3d(torch.zeros((2, 2, 2, 2)))


def test_vec_geom_matches_reference_formula():
    # tomobar/supp/funcs.py:45-81: ray Rz(t)(0,-1,0), det centre Rz(t)(c0,0,c1), u Rz(t)(1,0,0), v (0,0,1)
    from tomobar_amd.projector import geom_size, vec_geom_init3D
    th = np.array([0.0, 0.3, 1.7, 3.0])
    cor = np.array([[1.5, 0.0], [0.0, 0.0], [-2.0, 0.0], [0.25, 0.0]])
    v = vec_geom_init3D(th, 1.0, 1.0, cor)
    for i, t in enumerate(th):
        R = np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1.0]])
        np.testing.assert_allclose(v[i, 0:3], R @ [0, -1, 0], atol=1e-15)
        np.testing.assert_allclose(v[i, 3:6], R @ [cor[i, 0], 0, cor[i, 1]], atol=1e-15)
        np.testing.assert_allclose(v[i, 6:9], R @ [1, 0, 0], atol=1e-15)
        np.testing.assert_allclose(v[i, 9:12], [0, 0, 1], atol=1e-15)
    assert geom_size({"GridRowCount": 5, "GridColCount": 6, "GridSliceCount": 7}) == (7, 5, 6)
    assert geom_size({"DetectorRowCount": 3, "DetectorColCount": 9, "Vectors": v}) == (3, 4, 9)


def test_bench_ranks_take_arguments_from_environment(monkeypatch):
    """bench.py --gpus N re-executes itself under torch.distributed.run; the ranks must not receive the script's options
    on the launcher's command line (its parser rejects e.g. --n as an ambiguous prefix of its own options)."""
    import json
    import sys
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("TOMO_BENCH_ARGV", json.dumps(["--gpus", "2", "--n", "256", "--nz", "32", "--strong"]))
    a = bench.parse()
    assert (a.gpus, a.n, a.nz, a.strong) == (2, 256, 32, True)
    monkeypatch.delenv("RANK")
    assert bench.parse().gpus == 1   # a plain run ignores a stale variable


def test_bench_environment_overrides_and_north_star_footprint(monkeypatch):
    """A driver that only varies --gpus reaches the other workloads through BENCH_CONFIG / BENCH_STRONG; the north-star
    block (configs[4], strong scaling) is attempted only when a rank's share fits a 288 GB GPU: from 4 ranks on."""
    import sys
    import bench
    from tomobar_amd.slab import slab_bounds
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    a = bench.parse()
    assert (a.n, a.nz, a.angles, a.os, a.reg, a.strong) == (1024, 1024, 900, 12, "PD_TV", False) and not a.overridden
    monkeypatch.setenv("BENCH_CONFIG", "cfg5")
    monkeypatch.setenv("BENCH_STRONG", "1")
    a = bench.parse()
    assert (a.n, a.nz, a.angles, a.os, a.ring, a.strong) == (2560, 2160, 1800, 12, 1e-4, True) and not a.overridden
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "cfg3"])   # an explicit flag wins over the environment
    assert bench.parse().n == 2048
    monkeypatch.setenv("BENCH_CONFIG", "nonsense")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    with pytest.raises(SystemExit):
        bench.parse()
    monkeypatch.delenv("BENCH_CONFIG")
    monkeypatch.delenv("BENCH_STRONG")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "cfg5"])
    ns = bench.parse()
    fits = {}
    for world in (1, 2, 4, 8):
        share = max(slab_bounds(ns.nz, world, r)[1] - slab_bounds(ns.nz, world, r)[0] for r in range(world))
        fits[world] = bench.footprint_bytes(ns, share) < 0.9 * 288e9
    assert fits == {1: False, 2: False, 4: True, 8: True}, fits


def test_north_star_child_job_is_supervised(monkeypatch, tmp_path):
    """The north-star block of a multi-GPU run is a child job under a wall-clock limit: a child that hangs is killed with its
    whole process group and reported as skipped, a child that fails is reported with its exit code, a child that prints a
    bench line is merged -- in every case the caller (which has already persisted the headline) gets a block back."""
    import json
    import sys
    import time
    import bench
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    args = bench.parse()
    py = sys.executable
    # (1) hangs: killed after the limit, with the grandchild it spawned
    marker = str(tmp_path / "grandchild_alive")
    grand = tmp_path / "grand.py"
    grand.write_text(f"import time\ntime.sleep(4)\nopen({marker!r}, 'w').write('alive')\n")
    hang = tmp_path / "hang.py"
    hang.write_text(f"import subprocess, sys, time\nsubprocess.Popen([sys.executable, {str(grand)!r}])\ntime.sleep(60)\n")
    monkeypatch.setenv("BENCH_NORTH_STAR_CMD", json.dumps([py, str(hang)]))
    monkeypatch.setenv("BENCH_NORTH_STAR_TIMEOUT", "1.5")
    t0 = time.perf_counter()
    block = bench.north_star_child(args, 8, False, 288e9)
    assert time.perf_counter() - t0 < 20
    assert "killed after the" in block["skipped"] and "wall-clock limit" in block["skipped"], block
    assert block["per_gpu_slices"] == 270 and "value" not in block
    # (2) fails
    monkeypatch.setenv("BENCH_NORTH_STAR_TIMEOUT", "60")
    monkeypatch.setenv("BENCH_NORTH_STAR_CMD", json.dumps([py, "-c", "import sys; print('boom', file=sys.stderr); sys.exit(3)"]))
    block = bench.north_star_child(args, 8, False, 288e9)
    assert "exited with code 3" in block["skipped"] and "boom" in block["child_stderr_tail"], block
    # (3) succeeds: the child's JSON line (last line of its stdout that parses as a bench line) is merged; the child was told
    #     its workload through TOMO_BENCH_ARGV and that it is a child (no nesting)
    ok = ("import json, os; a = json.loads(os.environ['TOMO_BENCH_ARGV']); assert os.environ['TOMO_BENCH_CHILD'] == '1'; "
          "assert 'RANK' not in os.environ and 'BENCH_CONFIG' not in os.environ; print('noise'); "
          "print(json.dumps({'metric': 'fista_os_iterations_per_sec', 'value': 1.25, 'unit': 'iterations/s', 'n_gpus': 8, "
          "'scaling': 'strong', 'config': {'argv': a}, 'ms_per_step': 800.0, 'steps': 2, 'warmup': 1}))")
    monkeypatch.setenv("BENCH_NORTH_STAR_CMD", json.dumps([py, "-c", ok]))
    monkeypatch.setenv("RANK", "0")                 # what a rank under the driver's launcher sees; must not leak into the child
    monkeypatch.setenv("BENCH_CONFIG", "cfg2")
    block = bench.north_star_child(args, 8, False, 288e9)
    assert block["value"] == 1.25 and block["scaling"] == "strong" and "skipped" not in block, block
    argv = block["config"]["argv"]
    assert argv[:6] == ["--gpus", "8", "--config", "cfg5", "--strong", "--steps"] and "--no-north-star" in argv
    # (4) a share that cannot fit is not even started
    block = bench.north_star_child(args, 2, False, 288e9)
    assert "needs ~" in block["skipped"] and block["per_gpu_slices"] == 1080
    # the grandchild of (1) died with its group: it never wrote its marker
    time.sleep(max(0.0, 5.0 - (time.perf_counter() - t0)))
    assert not os.path.exists(marker)


def test_oracle_thread_team_follows_the_usable_cpus(oracle, monkeypatch):
    """The oracle's OpenMP team is sized to the CPUs the process may use (affinity mask capped by the cgroup quota), not
    to the logical CPU count of the host."""
    import os
    n = oracle.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if "OMP_NUM_THREADS" not in os.environ:
        assert oracle.threads() == n
    real_open = open

    def fake_open(path, *a, **k):   # a container that shows every CPU and grants two
        if path == "/sys/fs/cgroup/cpu.max":
            import io
            return io.StringIO("200000 100000\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    assert oracle.usable_cpus() == min(2, len(os.sched_getaffinity(0)))


def test_cupy_arrays_in_mean_cupy_arrays_out(monkeypatch):
    """VERDICT round 3, missing #5: the reference's classes return cupy.ndarray (methodsIR_CuPy.py:484).  CuPy cannot be
    installed in this image, so the plumbing is checked with a stand-in module that records the hand-over: a result is
    converted with cupy.from_dlpack exactly when the caller's array came from the cupy package, never otherwise."""
    import sys
    import types
    from tomobar_amd import ops

    fake = types.ModuleType("cupy")
    calls = []

    class ndarray:  # noqa: N801 -- the stand-in's arrays report the cupy package as their module
        pass
    ndarray.__module__ = "cupy"

    def from_dlpack(t):
        calls.append(t)
        a = ndarray()
        a.wrapped = t
        return a
    fake.ndarray, fake.from_dlpack = ndarray, from_dlpack
    monkeypatch.setitem(sys.modules, "cupy", fake)
    res = torch.zeros(3)
    given_cupy = ndarray()
    assert ops.is_cupy(given_cupy) and not ops.is_cupy(res) and not ops.is_cupy(np.zeros(2))
    out = ops.like(res, given_cupy)
    assert isinstance(out, ndarray) and out.wrapped is res and calls == [res]
    assert ops.like(res, res) is res and ops.like(res, np.zeros(2)) is res and ops.like(res, None) is res
    assert len(calls) == 1
    assert ops.base_ptr(res) == res.data_ptr() and ops.base_ptr(np.zeros(2)) is None
