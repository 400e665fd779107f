"""The measurement tooling that feeds bench.py's roofline block (no GPU needed): folding rocprofv3 --pmc CSVs into
the tracked summaries, and the names the engine, the profiler and the summaries use for one kernel."""

import csv
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, str(ROOT / "tools" / f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_profiler_names_map_to_engine_names():
    t = _tool("pmc_traffic")
    assert t.short_name("void k_gemm<1, 128, 0, 2, 1>(GemmArgs)")[0] == "k_gemm<f16_swish,128>"
    assert t.short_name("void k_gemm<4, 64, 4, 2, 1>(GemmArgs)") == ("k_gemm<resid,64>", "mubuf-register-staged loaders, int4 weights")
    assert t.short_name("void k_gemm256<1, 0>(GemmArgs)")[0] == "k_gemm256<f16_swish>"
    name, variant = t.short_name("void k_gemm256<3, 8>(GemmArgs)")
    assert name == "k_gemm256<glu>" and variant.endswith("int8 weights")
    assert t.short_name("void k_layernorm(float const*)") == (None, None)


def _write(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        w.writerows(rows)


def test_traffic_and_mfma_summaries(tmp_path, capsys):
    t = _tool("pmc_traffic")
    k = "void k_gemm256<1, 0>(GemmArgs)"
    _write(tmp_path / "fetch.csv", [{"Kernel_Name": k, "Counter_Name": "FETCH_SIZE", "Counter_Value": v} for v in (1000, 3000)])
    _write(tmp_path / "write.csv", [{"Kernel_Name": k, "Counter_Name": "WRITE_SIZE", "Counter_Value": v} for v in (500, 500)])
    t.traffic([str(tmp_path / "fetch.csv"), str(tmp_path / "write.csv"), str(tmp_path / "t.json"), "8064"])
    doc = json.loads((tmp_path / "t.json").read_text())
    row = doc["kernels"]["k_gemm256<f16_swish>"]
    # bytes = (2 x mean FETCH_SIZE + mean WRITE_SIZE) KiB: the gfx950 correction of the guide
    assert doc["rows"] == 8064 and row["hbm_bytes_per_launch"] == (2 * 2000 + 500) * 1024 and row["launches"] == 2
    rows = [{"Kernel_Name": k, "Counter_Name": n, "Counter_Value": v}
            for n, v in (("SQ_VALU_MFMA_BUSY_CYCLES", 16_000_000), ("SQ_BUSY_CYCLES", 1_250_000), ("SQ_WAVE_CYCLES", 1000),
                         ("SQ_WAIT_ANY", 300), ("SQ_WAIT_INST_ANY", 400), ("SQ_ACTIVE_INST_ANY", 300))]
    _write(tmp_path / "sq.csv", rows)
    t.mfma([str(tmp_path / "sq.csv"), str(tmp_path / "m.json"), "8064"])
    m = json.loads((tmp_path / "m.json").read_text())["kernels"]["k_gemm256<f16_swish>"]
    assert m["mfma_util"] == round(16_000_000 / (1_250_000 * 32), 4) and m["frac_wave_cycles_parked"] == 0.3
    capsys.readouterr()


def test_committed_summaries_name_the_kernel_the_bench_reports():
    """bench.py looks the replayed kernel's name up in the newest committed summaries; the FFN-up kernel at
    B = 64 x 10 s is the 256 x 256-tile one."""
    tr = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))[-1]
    mf = sorted((ROOT / "profiles").glob("r*_mfma_busy.json"))[-1]
    for f in (tr, mf):
        doc = json.loads(f.read_text())
        assert int(doc["rows"]) == 8064 and "k_gemm256<f16_swish>" in doc["kernels"], f.name
    assert 0.3 < json.loads(mf.read_text())["kernels"]["k_gemm256<f16_swish>"]["mfma_util"] < 1.0
