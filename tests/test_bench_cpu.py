"""CPU: bench.py's record helpers -- `roofline.traffic` is reported only for the workload and the kernels it was measured on."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _args(**kw):
    base = dict(dtype="bf16", d_model=1024, layers=12, opt=[])
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_kernel_set_hash_follows_the_step_kernels_sources():
    h = bench.kernel_set_hash()
    assert len(h) == 16 and int(h, 16) >= 0
    assert h == bench.kernel_set_hash()
    for name in bench.AR_STEP_KERNEL_SOURCES:
        assert os.path.exists(os.path.join(ROOT, "valle_amd", "csrc", name)), name


def test_kernel_set_hash_sees_code_not_comments():
    """The traffic record must survive a comment or white-space edit of the kernel sources and go stale with any change of code."""
    a = "int f(int x) {  // doubles\n  return 2 * x; /* twice\n   as much */\n}\n"
    b = "int f(int x) {\n\n    return 2 * x;\n}  // same function, other words\n"
    c = "int f(int x) {\n  return 3 * x;\n}\n"
    assert bench._code_only(a) == bench._code_only(b) != bench._code_only(c)
    assert "//" not in bench._code_only(a) and "/*" not in bench._code_only(a)


def test_measured_traffic_is_gated_on_workload_and_kernel_set(tmp_path, monkeypatch):
    # other workloads never get the C2 batch-1 figure
    for a, B in ((_args(dtype="fp8w"), 1), (_args(), 64), (_args(d_model=1536), 1), (_args(opt=["qkv_attn=0"]), 1)):
        v, why = bench.measured_traffic(a, B)
        assert v is None and "default workload" in why
    # the default workload: the committed file's figure while its hash matches, else null + "stale"
    f = tmp_path / "traffic.json"
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    v, why = bench.measured_traffic(_args(), 1)
    assert v is None and "missing" in why
    f.write_text(json.dumps({"kernel_set": bench.kernel_set_hash(), "bytes_per_step": 354185471, "source": "profiles/x.csv"}))
    v, why = bench.measured_traffic(_args(), 1)
    assert v == 354185471 and why == "profiles/x.csv"
    f.write_text(json.dumps({"kernel_set": "0" * 16, "bytes_per_step": 1}))
    v, why = bench.measured_traffic(_args(), 1)
    assert v is None and why.startswith("stale")


def test_committed_profile_files_parse():
    with open(os.path.join(ROOT, "profiles", "ar_step_traffic.json")) as f:
        t = json.load(f)
    assert t["bytes_per_step"] > 300e6 and len(t["kernel_set"]) == 16
    with open(os.path.join(ROOT, "profiles", "cpu_baseline_n1.json")) as f:
        c = json.load(f)
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
