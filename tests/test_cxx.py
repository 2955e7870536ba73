"""C++ side of the boundary: the plugin headers under include/ compile as downstream code uses
them, their host logic passes on CPU, and (gpu) the virtual API matches the oracle on the device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cxx", "build")
LIBDIR = os.path.join(ROOT, "common_amd", "lib")
LINK = ["-L" + LIBDIR, "-lmicroscopes_hip", "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-lamdhip64",
        "-Wl,-rpath,/opt/rocm/lib"]


def _audited(output):
    """the C++ tests print one "AUDIT name max_error gate checks" line per gate (tests/cxx/audit.hpp): folded into the
    session's tolerance audit beside the Python gates (tests/gpu_helpers.py)"""
    from tests.gpu_helpers import audit
    n = 0
    for line in output.splitlines():
        if line.startswith("AUDIT "):
            _, name, err, gate, _checks = line.split()
            audit("cxx." + name, float(err), float(gate))
            n += 1
    return n


def _cxx(src, out, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, out)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-o", exe, src] + list(extra))
    return exe


def test_host_api_cpu():
    exe = _cxx(os.path.join(ROOT, "tests", "cxx", "test_host_api.cpp"), "test_host_api")
    assert "test_host_api ok" in subprocess.check_output([exe]).decode()


def test_plugin_layer_writes_protobufs_own_bytes_for_the_in_tree_messages(golden):
    """get_hp / get_ss of bbnc and dm, group_manager::get_hp and ::serialize (the reference's own
    test_group_manager.cpp:22-66 scenario) against tests/golden/wire.json -- bytes from Google's protobuf runtime
    (microscopes/io/schema.proto:3-46).  Host-side only."""
    exe = _cxx(os.path.join(ROOT, "tests", "cxx", "test_wire_golden.cpp"), "test_wire_golden", LINK)
    got = {}
    for line in subprocess.check_output([exe]).decode().splitlines():
        name, idx, hx = line.split("\t")
        got.setdefault(name, {})[int(idx)] = hx
    want = {}
    for vec in golden("wire"):
        want.setdefault(vec["message"], []).append(vec["hex"])
    for name in ("CRP", "BetaBernoulliNonConj.Shared", "BetaBernoulliNonConj.Group", "DirichletMultinomial.Shared",
                 "DirichletMultinomial.Group", "GroupManager"):
        assert [got[name][i] for i in range(len(want[name]))] == want[name], name


def test_library_exports_every_symbol_the_header_declares():
    import re
    import common_amd
    hdr = open(os.path.join(ROOT, "include", "microscopes_hip.h")).read()
    declared = set(re.findall(r"\b(msc_[a-z0-9_]+)\s*\(", hdr))
    lib = common_amd.load()          # dlopen only; no device call
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(common_amd.EXPORTS), declared ^ set(common_amd.EXPORTS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", common_amd.LIB_PATH]).decode()
    exported = set(re.findall(r" T (msc_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under common_amd/, include/, bin/ may use it"""
    bad = []
    for base in ("common_amd", "include", "bin"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp", ".inc")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if "msc_oracle" in txt or "from oracle" in txt or "import oracle" in txt or "orc_f" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_device_fails_loudly():
    import torch
    import common_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(common_amd.MicroscopesHipError):
        common_amd.Context()


@pytest.mark.skipif(not os.path.exists("/root/reference/bin/perf_group.cpp"),
                    reason="reference tree not mounted (GPU box)")
def test_reference_perf_group_compiles_unchanged_against_our_headers():
    """drop-in check of the C++ surface: the reference's own microbenchmark source, compiled from
    where it lies (never copied), builds and links against include/ + the HIP library"""
    _cxx("/root/reference/bin/perf_group.cpp", "perf_group_reference_source", LINK)


@pytest.mark.gpu
def test_plugin_api_on_device_matches_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libmsc_oracle.so"])
    exe = _cxx(os.path.join(ROOT, "tests", "cxx", "test_plugin_gpu.cpp"), "test_plugin_gpu",
               LINK + ["-L" + os.path.join(ROOT, "oracle"), "-lmsc_oracle",
                       "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.check_output([exe]).decode()
    assert "test_plugin_gpu ok" in out and _audited(out) >= 10


@pytest.mark.gpu
def test_relation_dataviews_on_device():
    exe = _cxx(os.path.join(ROOT, "tests", "cxx", "test_relation_gpu.cpp"), "test_relation_gpu",
               ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"] + LINK)
    out = subprocess.check_output([exe]).decode()
    assert "test_relation_gpu ok" in out and _audited(out) >= 1


@pytest.mark.gpu
def test_c_level_collective_over_rccl():
    """msc_comm_* / msc_state_allreduce / msc_sweep_step_sharded from a plain C++ host: RCCL on the box's GPU"""
    exe = _cxx(os.path.join(ROOT, "tests", "cxx", "test_comm_gpu.cpp"), "test_comm_gpu",
               ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"] + LINK)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    assert "test_comm_gpu ok" in subprocess.check_output([exe], env=env, timeout=300).decode()


@pytest.mark.gpu
def test_perf_group_harness_runs():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bin")])
    out = subprocess.check_output([os.path.join(ROOT, "bin", "perf_group_hip"), "64", "2"]).decode()
    assert "noop virtual API" in out and "bb batched C ABI" in out and "score_value evals/s" in out


def test_mixture_state_builds_against_the_reference_interface_names():
    """the entity_based_state_object implementation compiles and links as downstream code would use it (no device call)"""
    _cxx(os.path.join(ROOT, "tests", "cxx", "test_mixture_state_gpu.cpp"), "test_mixture_state_gpu", LINK)


@pytest.mark.gpu
def test_mixture_state_per_entity_gibbs_and_batched_sweep():
    exe = _cxx(os.path.join(ROOT, "tests", "cxx", "test_mixture_state_gpu.cpp"), "test_mixture_state_gpu", LINK)
    out = subprocess.check_output([exe]).decode()
    assert "test_mixture_state_gpu ok" in out and _audited(out) >= 8


def test_sample_value_draws_from_the_posterior_predictive():
    """host-side samplers of the device-backed models (no device call): moments against the closed forms"""
    exe = _cxx(os.path.join(ROOT, "tests", "cxx", "test_sample_value.cpp"), "test_sample_value", LINK)
    assert "test_sample_value ok" in subprocess.check_output([exe]).decode()
